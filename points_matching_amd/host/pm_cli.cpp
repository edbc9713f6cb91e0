// pm_cli — C++ host of the MI355X-native matcher: the counterpart of the reference's main()
// (`Points Matching/main.cpp:9-147`) for the hot path main.cpp:42-123.  It keeps the reference's
// stdout surface (main.cpp:58-59, :73, :76, :119, :123) and calls the HIP kernels only through
// the C ABI of include/pm.h.  Images come as binary PGM/PPM (no JPEG/BMP decoder in this image: `convert a.jpg a.pgm`
// or PIL does it) and go through the build-owned detector/descriptor of pm_features.cpp (the
// reference's SURF, main.cpp:22-40, is OpenCV nonfree code); descriptors and keypoints can also come from matrix files.
//
//   pm_cli --img1 left.pgm --img2 right.pgm [--max-kp 4000]        (image pair in: main.cpp:14-15, :22-40)
//   pm_cli --desc1 a.pmm --desc2 b.pmm --kp1 ka.pmm --kp2 kb.pmm   (descriptor pair in)
//          [--filter midpoint|ratio] [--ratio 0.8] [--iters 10000] [--thresh 1.0] [--seed 24301]
//          [--method 7point-lmeds|ransac8] [--f-scale opencv|unit] [--device 0] [--gpus N] [--quiet] [--json]
//          [--print-epilines] [--epilines out.ppm [--canvas W H] [--img2 right.pgm]] [--matcher bf|flann]
//          [--knn-hint auto|int|u8|unit]   what the caller knows about float descriptors (pm.h PM_KNN_HINT_*; default auto, and
//                                     u8 for --img1/--img2, whose extractor writes u8-valued rows): route only, same output
// --matcher flann: the reference's ACTIVE matcher object (`FlannBasedMatcher matcher;`, main.cpp:44): 4 randomised
// kd-trees, 32 checks (pm_flann_*; approximate, seeded by --seed); bf (default) is the exact matcher of main.cpp:43.
// NOTE: the default matcher DEVIATES from main.cpp:44 on purpose — the exact matcher is faster on this hardware at every
// size and reproducible (FLANN seeds its trees from C rand()); `--matcher flann` gives the reference's literal flow
// (FLANN 1-NN -> midpoint filter -> 7-point LMedS).  The image front end (--img1/--img2) is a SIFT-style detector, not
// the reference's SURF(8000) (main.cpp:22-40): no compatibility with OpenCV's keypoints is claimed.
// --gpus N (> 1): matcher rows and hypothesis ids are sharded over N GPUs through pm_mgpu_match_ransac (RCCL behind
// the C ABI); needs --filter ratio --method ransac8 (the sharded form of the path, BASELINE config C4).
// --print-epilines / --epilines: main.cpp:127-142 — the epipolar lines of the image-1 points in image 2
// (computeCorrespondEpilines) and the two end points main.cpp:138-140 hands to cv::line; --epilines draws them in
// white over --img2 (binary PGM, as main.cpp:137 draws over img2) or over a black canvas and writes a PPM.
// --method 7point-lmeds (default) is the estimator the reference's CV_FM_7POINT call literally selects (7-point
// solver inside the LMedS loop, OpenCV's 300 iterations unless --iters is given); ransac8 is the estimator
// BASELINE.json names (normalised 8-point + Sampson, --iters defaults to 10000).
//
// .pmm file: magic "PMM1", int32 rows, int32 cols, int32 dtype (0 = float32, 1 = uint8), data
// row-major.  float32 descriptors -> BF-L2 (main.cpp:43), uint8 -> BF-Hamming.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "pm.h"
#include "pm_features.hpp"

namespace {

struct Matrix {
    int rows = 0, cols = 0, dtype = 0;
    std::vector<unsigned char> data;
    const float* f32() const { return reinterpret_cast<const float*>(data.data()); }
};

bool load_matrix(const std::string& path, Matrix& m)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "pm_cli: cannot open %s\n", path.c_str()); return false; }
    char magic[4];
    int32_t hdr[3];
    bool ok = fread(magic, 1, 4, f) == 4 && memcmp(magic, "PMM1", 4) == 0 && fread(hdr, 4, 3, f) == 3;
    if (ok) {
        m.rows = hdr[0]; m.cols = hdr[1]; m.dtype = hdr[2];
        ok = m.rows >= 0 && m.cols > 0 && (m.dtype == 0 || m.dtype == 1);
    }
    if (ok) {
        const size_t bytes = static_cast<size_t>(m.rows) * m.cols * (m.dtype == 0 ? 4 : 1);
        m.data.resize(bytes);
        ok = fread(m.data.data(), 1, bytes, f) == bytes;
    }
    fclose(f);
    if (!ok) fprintf(stderr, "pm_cli: %s is not a valid .pmm matrix\n", path.c_str());
    return ok;
}

// binary PGM (P5, maxval <= 255): what img2 of main.cpp:21 would be after a grey conversion
bool load_pgm(const std::string& path, std::vector<unsigned char>& px, int& w, int& h)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "pm_cli: cannot open %s\n", path.c_str()); return false; }
    auto token = [&](int& out) -> bool {
        int c = fgetc(f);
        for (;;) {
            while (c == ' ' || c == '\n' || c == '\r' || c == '\t') c = fgetc(f);
            if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
            break;
        }
        if (c < '0' || c > '9') return false;
        long v = 0;
        while (c >= '0' && c <= '9') { v = v * 10 + (c - '0'); if (v > 1000000) return false; c = fgetc(f); }
        out = static_cast<int>(v);
        return true;                                 // the single whitespace after the token has been consumed
    };
    char magic[2] = {0, 0};
    int maxv = 0;
    bool ok = fread(magic, 1, 2, f) == 2 && magic[0] == 'P' && magic[1] == '5' && token(w) && token(h) && token(maxv) &&
              w > 0 && h > 0 && maxv > 0 && maxv <= 255;
    if (ok) {
        px.resize(static_cast<size_t>(w) * h);
        ok = fread(px.data(), 1, px.size(), f) == px.size();
    }
    fclose(f);
    if (!ok) fprintf(stderr, "pm_cli: %s is not a binary 8-bit PGM\n", path.c_str());
    return ok;
}

// cv::line(img2, p1, p2, Scalar(255,255,255)) for every epipolar line (main.cpp:134-141): 8-connected Bresenham
// walk from p1 to p2, pixels outside the image skipped.  Written as a binary PPM (P6).
bool write_epiline_ppm(const std::string& path, const std::vector<unsigned char>& gray, int w, int h,
                       const std::vector<int32_t>& ends, int n)
{
    std::vector<unsigned char> rgb(static_cast<size_t>(w) * h * 3, 0);
    if (!gray.empty())
        for (size_t i = 0; i < static_cast<size_t>(w) * h; ++i) rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = gray[i];
    for (int i = 0; i < n; ++i) {
        long long x0 = ends[4 * i], y0 = ends[4 * i + 1], x1 = ends[4 * i + 2], y1 = ends[4 * i + 3];
        if (y0 == INT32_MIN || y1 == INT32_MIN) continue;              // vertical line (b == 0): no finite end points
        // clip the y range coarsely first: a nearly vertical line may have end points billions of pixels away
        const long long dx = x1 - x0, dy = y1 - y0;
        const long long adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
        const long long steps = adx > ady ? adx : ady;
        if (steps == 0) { if (x0 >= 0 && x0 < w && y0 >= 0 && y0 < h) { size_t o = 3 * (static_cast<size_t>(y0) * w + x0); rgb[o] = rgb[o + 1] = rgb[o + 2] = 255; } continue; }
        if (ady <= adx) {                                               // x-major: one pixel per column
            for (long long x = (x0 < 0 ? 0 : x0); x <= (x1 >= w ? w - 1 : x1); ++x) {
                const long long y = y0 + (2 * (x - x0) * dy + (dy >= 0 ? dx : -dx)) / (2 * dx);   // rounded, dx > 0 here
                if (y >= 0 && y < h) { size_t o = 3 * (static_cast<size_t>(y) * w + x); rgb[o] = rgb[o + 1] = rgb[o + 2] = 255; }
            }
        } else {                                                        // y-major: one pixel per row
            const long long ya = y0 < y1 ? y0 : y1, yb = y0 < y1 ? y1 : y0;
            for (long long y = (ya < 0 ? 0 : ya); y <= (yb >= h ? h - 1 : yb); ++y) {
                const long long num = 2 * (y - y0) * dx + (((y - y0) * dx >= 0) == (dy >= 0) ? dy : -dy);
                const long long x = x0 + num / (2 * dy);
                if (x >= 0 && x < w) { size_t o = 3 * (static_cast<size_t>(y) * w + x); rgb[o] = rgb[o + 1] = rgb[o + 2] = 255; }
            }
        }
    }
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "pm_cli: cannot write %s\n", path.c_str()); return false; }
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    const bool ok = fwrite(rgb.data(), 1, rgb.size(), f) == rgb.size();
    fclose(f);
    return ok;
}

bool save_matrix(const std::string& path, const Matrix& m)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "pm_cli: cannot write %s\n", path.c_str()); return false; }
    const int32_t hdr[3] = {m.rows, m.cols, m.dtype};
    const bool ok = fwrite("PMM1", 1, 4, f) == 4 && fwrite(hdr, 4, 3, f) == 3 && fwrite(m.data.data(), 1, m.data.size(), f) == m.data.size();
    fclose(f);
    return ok;
}

int fail(const char* what, int rc)
{
    fprintf(stderr, "pm_cli: %s failed: %s (%s)\n", what, pm_status_string(rc), pm_last_error());
    return 1;
}

}  // namespace

int main(int argc, char** argv)
{
    std::string desc1, desc2, kp1, kp2, filter = "midpoint", fscale = "opencv", method = "7point-lmeds";
    float ratio = 0.8f, thresh = 1.0f;
    long iters = 10000;
    unsigned long long seed = 0x5EED;
    int device = 0, gpus = 1, canvas_w = 993, canvas_h = 660;       // canvas default: the size of img01/img02
    std::string epi_ppm, img2_path, matcher = "bf", img1_path, save_prefix, knn_hint;
    bool extract_only = false;
    int max_kp = 4000;
    bool quiet = false, json = false, iters_given = false, print_epi = false, force_mgpu = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&](const char* name) -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "pm_cli: %s needs a value\n", name); exit(2); }
            return argv[++i];
        };
        if (a == "--desc1") desc1 = val("--desc1");
        else if (a == "--desc2") desc2 = val("--desc2");
        else if (a == "--kp1") kp1 = val("--kp1");
        else if (a == "--kp2") kp2 = val("--kp2");
        else if (a == "--filter") filter = val("--filter");
        else if (a == "--ratio") ratio = strtof(val("--ratio"), nullptr);
        else if (a == "--iters") { iters = strtol(val("--iters"), nullptr, 0); iters_given = true; }
        else if (a == "--thresh") thresh = strtof(val("--thresh"), nullptr);
        else if (a == "--seed") seed = strtoull(val("--seed"), nullptr, 0);
        else if (a == "--f-scale") fscale = val("--f-scale");
        else if (a == "--method") method = val("--method");
        else if (a == "--device") device = atoi(val("--device"));
        else if (a == "--gpus") gpus = atoi(val("--gpus"));
        else if (a == "--matcher") matcher = val("--matcher");
        else if (a == "--knn-hint") knn_hint = val("--knn-hint");      // auto | int | u8: what the caller knows about float descriptors
        else if (a == "--mgpu") force_mgpu = true;            // take the pm_mgpu path even with --gpus 1 (tests)
        else if (a == "--print-epilines") print_epi = true;
        else if (a == "--epilines") epi_ppm = val("--epilines");
        else if (a == "--img2") img2_path = val("--img2");
        else if (a == "--img1") img1_path = val("--img1");
        else if (a == "--max-kp") max_kp = atoi(val("--max-kp"));
        else if (a == "--save-features") save_prefix = val("--save-features");   // PREFIX_{desc1,desc2,kp1,kp2}.pmm
        else if (a == "--extract-only") extract_only = true;                     // stop after the feature front-end (no GPU needed)
        else if (a == "--canvas") { canvas_w = atoi(val("--canvas")); canvas_h = atoi(val("--canvas")); }
        else if (a == "--quiet") quiet = true;
        else if (a == "--json") json = true;
        else { fprintf(stderr, "pm_cli: unknown option %s\n", a.c_str()); return 2; }
    }
    Matrix d1, d2, k1, k2;
    const bool from_images = !img1_path.empty();
    // what the caller knows about float descriptors (pm.h: a hint is verified on the device, a wrong one only costs time).
    // The build-owned extractor of --img1/--img2 writes u8-valued rows, so that path says so unless told otherwise.
    if (knn_hint.empty()) knn_hint = from_images ? "u8" : "auto";
    if (knn_hint != "auto" && knn_hint != "int" && knn_hint != "u8" && knn_hint != "unit") {
        fprintf(stderr, "pm_cli: --knn-hint auto|int|u8|unit\n");
        return 2;
    }
    const int knn_flags = knn_hint == "u8" ? PM_KNN_HINT_U8 : knn_hint == "int" ? PM_KNN_HINT_INTEGER :
                          knn_hint == "unit" ? PM_KNN_HINT_UNIT_NORM : 0;          // unit: L2-normalised rows (SURF, main.cpp:37-40)
    if (from_images) {
        // ---- imread + detect + compute                                          main.cpp:14-15, :22-26, :36-40
        if (img2_path.empty() || max_kp < 8) { fprintf(stderr, "pm_cli: --img1 needs --img2 (and --max-kp >= 8)\n"); return 2; }
        const std::string* paths[2] = {&img1_path, &img2_path};
        Matrix* dm[2] = {&d1, &d2};
        Matrix* km[2] = {&k1, &k2};
        for (int i = 0; i < 2; ++i) {
            pm_feat::Image im;
            std::string err;
            if (!pm_feat::load_pnm_gray(*paths[i], im, err)) { fprintf(stderr, "pm_cli: %s\n", err.c_str()); return 1; }
            const pm_feat::Features ft = pm_feat::detect_and_describe(im, max_kp);
            dm[i]->rows = ft.n; dm[i]->cols = 128; dm[i]->dtype = 0;
            dm[i]->data.assign(reinterpret_cast<const unsigned char*>(ft.desc.data()),
                               reinterpret_cast<const unsigned char*>(ft.desc.data()) + ft.desc.size() * sizeof(float));
            km[i]->rows = ft.n; km[i]->cols = 2; km[i]->dtype = 0;
            km[i]->data.assign(reinterpret_cast<const unsigned char*>(ft.kp_xy.data()),
                               reinterpret_cast<const unsigned char*>(ft.kp_xy.data()) + ft.kp_xy.size() * sizeof(float));
            if (!quiet) fprintf(stderr, "pm_cli: %s: %d x %d, %d keypoints\n", paths[i]->c_str(), im.w, im.h, ft.n);
            if (i == 1) { canvas_w = im.w; canvas_h = im.h; }
        }
        if (!save_prefix.empty()) {
            const Matrix* ms[4] = {&d1, &d2, &k1, &k2};
            const char* names[4] = {"_desc1.pmm", "_desc2.pmm", "_kp1.pmm", "_kp2.pmm"};
            for (int i = 0; i < 4; ++i)
                if (!save_matrix(save_prefix + names[i], *ms[i])) return 1;
        }
        if (extract_only) return 0;
        if (d1.rows < 8 || d2.rows < 8) { fprintf(stderr, "pm_cli: too few keypoints\n"); return 1; }
    } else {
    if (desc1.empty() || desc2.empty() || kp1.empty() || kp2.empty()) {
        fprintf(stderr, "usage: pm_cli (--img1 L.pgm --img2 R.pgm | --desc1 A --desc2 B --kp1 KA --kp2 KB) [--filter midpoint|ratio] "
                        "[--ratio r] [--method 7point-lmeds|ransac8] [--iters n] [--thresh px] [--seed s] [--f-scale opencv|unit] "
                        "[--matcher bf|flann] [--knn-hint auto|int|u8|unit] [--gpus N] [--print-epilines] [--epilines out.ppm] [--json] [--quiet]\n"
                        "  (default matcher bf = exact brute force, main.cpp:43; the reference's active one is --matcher flann, main.cpp:44)\n");
        return 2;
    }
    if (!load_matrix(desc1, d1) || !load_matrix(desc2, d2) || !load_matrix(kp1, k1) || !load_matrix(kp2, k2)) return 1;
    if (d1.cols != d2.cols || d1.dtype != d2.dtype || k1.dtype != 0 || k2.dtype != 0 || k1.cols != 2 ||
        k2.cols != 2 || k1.rows != d1.rows || k2.rows != d2.rows) {
        fprintf(stderr, "pm_cli: inconsistent descriptor / keypoint matrices\n");
        return 1;
    }
    }
    const bool want_ratio = filter == "ratio";
    if (method != "ransac8" && method != "7point-lmeds") { fprintf(stderr, "pm_cli: --method ransac8|7point-lmeds\n"); return 2; }
    if (!want_ratio && filter != "midpoint") { fprintf(stderr, "pm_cli: --filter midpoint|ratio\n"); return 2; }
    if (method == "ransac8" && !iters_given) iters = 10000;
    if ((gpus > 1 || force_mgpu) && (!want_ratio || method != "ransac8")) {
        fprintf(stderr, "pm_cli: --gpus N needs --filter ratio --method ransac8\n");
        return 2;
    }
    if (gpus < 1 || canvas_w < 1 || canvas_h < 1) { fprintf(stderr, "pm_cli: bad --gpus / --canvas\n"); return 2; }
    if (matcher != "bf" && matcher != "flann") { fprintf(stderr, "pm_cli: --matcher bf|flann\n"); return 2; }
    if (matcher == "flann" && (d1.dtype != 0 || gpus > 1 || force_mgpu)) {
        fprintf(stderr, "pm_cli: --matcher flann needs float32 descriptors and a single GPU\n");
        return 2;
    }

    using clk = std::chrono::steady_clock;
    pm_ctx* ctx = nullptr;
    pm_mgpu* mg = nullptr;
    int rc = PM_OK, est_rc = PM_OK;
    std::vector<pm_match> good(static_cast<size_t>(d1.rows) + 1);
    int n_good = 0, n_inl = 0;
    std::vector<float> xy1, xy2;
    std::vector<uint8_t> mask(static_cast<size_t>(d1.rows) + 1);
    double F[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t key = 0;
    long long lmeds_model = -1;
    pm_ransac_params prm;
    prm.hyp_begin = 0; prm.hyp_end = iters; prm.seed = seed; prm.thresh_px = thresh; prm.error_kind = PM_ERR_SAMPSON;
    auto t0 = clk::now(), t1 = t0, t2 = t0, t3 = t0;
    auto gather_and_list = [&]() -> int {
        // ---- match list + index vectors + KeyPoint::convert                    main.cpp:71-79, 89-91
        std::vector<int32_t> qi(n_good), ti(n_good);
        xy1.assign(2 * static_cast<size_t>(n_good), 0.f);
        xy2.assign(2 * static_cast<size_t>(n_good), 0.f);
        int r = pm_match_indices(good.data(), n_good, qi.data(), ti.data());
        if (r == PM_OK) r = pm_gather_points(k1.f32(), k1.rows, qi.data(), n_good, xy1.data());
        if (r == PM_OK) r = pm_gather_points(k2.f32(), k2.rows, ti.data(), n_good, xy2.data());
        if (r != PM_OK) return r;
        if (!quiet) {
            const long need = pm_format_match_list(good.data(), n_good, nullptr, 0);
            std::string text(static_cast<size_t>(need) + 1, '\0');
            pm_format_match_list(good.data(), n_good, &text[0], text.size());
            fputs(text.c_str(), stdout);
        }
        return PM_OK;
    };
    if (gpus > 1 || force_mgpu) {
        // ---- the sharded form of main.cpp:46 .. :98 (query rows + hypothesis ids over `gpus` devices)
        rc = pm_mgpu_create(gpus, nullptr, &mg);
        if (rc != PM_OK) return fail("pm_mgpu_create", rc);
        t0 = clk::now();
        rc = pm_mgpu_match_ransac(mg, d1.data.data(), d1.rows, d2.data.data(), d2.rows, d1.cols, d1.dtype == 1, k1.f32(), k2.f32(),
                                  ratio, knn_flags, &prm, good.data(), &n_good, F, mask.data(), &n_inl, &key);
        if (rc != PM_OK && rc != PM_E_TOO_FEW && rc != PM_E_NO_MODEL) return fail("pm_mgpu_match_ransac", rc);
        est_rc = rc;
        t1 = t2 = t3 = clk::now();
        good.resize(n_good);
        const int r2 = gather_and_list();
        if (r2 != PM_OK) return fail("gather", r2);
    } else {
    rc = pm_ctx_create(device, &ctx);
    if (rc != PM_OK) return fail("pm_ctx_create", rc);
    t0 = clk::now();

    // ---- matcher.match(imageDesc1, imageDesc2, matchePoints, Mat())            main.cpp:42-46
    const int k = want_ratio ? 2 : 1;
    std::vector<pm_match> knn(static_cast<size_t>(d1.rows) * k);
    if (matcher == "flann") {
        pm_flann_params fp;
        fp.trees = 4; fp.checks = 32; fp.seed = seed;               // cv::flann defaults behind main.cpp:44
        pm_flann_index* ix = nullptr;
        rc = pm_flann_build(ctx, d2.f32(), d2.rows, d2.cols, &fp, &ix);
        if (rc == PM_OK) rc = pm_flann_knn_l2_f32(ctx, ix, d1.f32(), d1.rows, k, knn.data());
        pm_flann_destroy(ix);
    } else if (d1.dtype == 0)
        rc = pm_bf_knn_l2_f32(ctx, d1.f32(), d1.rows, d2.f32(), d2.rows, d1.cols, k, knn_flags, knn.data());
    else
        rc = pm_bf_knn_hamming_u8(ctx, d1.data.data(), d1.rows, d2.data.data(), d2.rows, d1.cols, k, knn.data());
    if (rc != PM_OK) return fail("matcher", rc);
    t1 = clk::now();

    // ---- selecting strong features                                             main.cpp:48-69
    if (want_ratio) {
        rc = pm_filter_ratio(knn.data(), d1.rows, k, ratio, good.data(), &n_good);
    } else {
        // OpenCV drops queries without a neighbour (empty train set): keep only matched rows
        std::vector<pm_match> matched;
        for (const pm_match& m : knn) if (m.trainIdx >= 0) matched.push_back(m);
        double lo = 0, hi = 0;
        rc = pm_filter_midpoint(matched.data(), static_cast<int>(matched.size()), &lo, &hi, good.data(), &n_good);
        if (rc == PM_OK && !quiet) {
            // main.cpp:58-59; the full-width colon of the source is not representable in the
            // reference's code page and its binary prints '?' 
            std::cout << "The Best Match is? " << lo << std::endl;
            std::cout << "The Worst Match is? " << hi << std::endl;
        }
    }
    if (rc != PM_OK) return fail("filter", rc);
    good.resize(n_good);
    rc = gather_and_list();
    if (rc != PM_OK) return fail("gather", rc);
    t2 = clk::now();

    // ---- cv::findFundamentalMat(...)                                           main.cpp:94-98
    if (method == "7point-lmeds") {
        pm_lmeds_params lp;
        lp.hyp_begin = 0; lp.hyp_end = iters_given ? iters : pm_lmeds_default_iters(0.99, 0.45); lp.seed = seed;
        int64_t best_model = -1;
        double median = 0;
        rc = pm_lmeds_fundamental(ctx, xy1.data(), xy2.data(), n_good, &lp, F, mask.data(), &n_inl, &best_model, &median);
        lmeds_model = best_model;
        if (rc != PM_OK && rc != PM_E_TOO_FEW && rc != PM_E_NO_MODEL) return fail("pm_lmeds_fundamental", rc);
    } else {
        rc = pm_ransac_fundamental(ctx, xy1.data(), xy2.data(), n_good, &prm, F, mask.data(), &n_inl, &key);
        if (rc != PM_OK && rc != PM_E_TOO_FEW && rc != PM_E_NO_MODEL) return fail("pm_ransac_fundamental", rc);
    }
    est_rc = rc;                             // the estimator's own status (PM_E_TOO_FEW / PM_E_NO_MODEL are results, not failures)
    t3 = clk::now();
    }
    // like cv::findFundamentalMat, a failed estimate yields the zero matrix
    if (fscale == "opencv") pm_f_scale_f33(F);

    // ---- residual report, the reference's x1^T F x2 form                       main.cpp:101-123
    std::vector<double> res(static_cast<size_t>(n_good) + 1);
    double mean_abs = 0;
    rc = pm_epipolar_residuals(xy1.data(), xy2.data(), n_good, F, /*transposed=*/1, res.data(), &mean_abs);
    if (rc != PM_OK) return fail("pm_epipolar_residuals", rc);
    if (!quiet) {
        for (int i = 0; i < n_good; ++i) std::cout << "result = " << i << " " << res[i] << std::endl;   // :119
        std::cout << "The average value is  " << mean_abs << std::endl;                                 // :123
    }
    // ---- computeCorrespondEpilines(selPoints1, 1, F, lines1) + the end points of cv::line      main.cpp:127-142
    if (print_epi || !epi_ppm.empty()) {
        int cols = canvas_w, rows_ = canvas_h;
        std::vector<unsigned char> gray;
        if (!img2_path.empty() && !load_pgm(img2_path, gray, cols, rows_)) return 1;
        std::vector<float> lines(3 * static_cast<size_t>(n_good) + 3);
        std::vector<int32_t> ends(4 * static_cast<size_t>(n_good) + 4);
        rc = pm_epilines(xy1.data(), n_good, 1, F, lines.data());
        if (rc == PM_OK) rc = pm_epiline_endpoints(lines.data(), n_good, cols, ends.data());
        if (rc != PM_OK) return fail("pm_epilines", rc);
        if (print_epi)
            for (int i = 0; i < n_good; ++i)
                printf("epiline = %d %g %g %g  (%d, %d) -> (%d, %d)\n", i, lines[3 * i], lines[3 * i + 1], lines[3 * i + 2],
                       ends[4 * i], ends[4 * i + 1], ends[4 * i + 2], ends[4 * i + 3]);
        if (!epi_ppm.empty() && !write_epiline_ppm(epi_ppm, gray, cols, rows_, ends, n_good)) return 1;
    }
    if (json) {
        auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        double mean_fwd = 0;
        pm_epipolar_residuals(xy1.data(), xy2.data(), n_good, F, 0, nullptr, &mean_fwd);
        printf("{\"n1\": %d, \"n2\": %d, \"dim\": %d, \"matches\": %d, \"inliers\": %d, \"best_hyp\": %lld, "
               "\"ransac_status\": %d, \"mean_abs_x1Fx2\": %.17g, \"mean_abs_x2Fx1\": %.17g, "
               "\"ms\": {\"match\": %.3f, \"filter_gather\": %.3f, \"ransac\": %.3f}, "
               "\"F\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g]}\n",
               d1.rows, d2.rows, d1.cols, n_good, n_inl,
               method == "7point-lmeds" ? lmeds_model : static_cast<long long>(key ? pm_ransac_key_hyp(key) : 0u), est_rc,
               mean_abs, mean_fwd, ms(t0, t1), ms(t1, t2), ms(t2, t3), F[0], F[1], F[2], F[3], F[4], F[5], F[6],
               F[7], F[8]);
    }
    pm_ctx_destroy(ctx);
    pm_mgpu_destroy(mg);
    return 0;
}
