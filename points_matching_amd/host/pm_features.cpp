// pm_features.cpp — see pm_features.hpp.  Slot in the reference: main.cpp:22-26, :36-40 (feature front-end).
#include "pm_features.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace pm_feat {
namespace {

struct Plane {
    int w = 0, h = 0;
    std::vector<float> v;
    float at(int y, int x) const { return v[static_cast<size_t>(y) * w + x]; }
};

int reflect(int i, int n)                     // half-sample symmetric: ... 1 0 | 0 1 2 ... n-1 | n-1 n-2 ...
{
    while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - 1 - i;
    return i;
}

// separable Gaussian, taps out to 4 sigma, double accumulation, rows (y) first then columns (x)
Plane gaussian(const Plane& in, double sigma)
{
    const int r = static_cast<int>(4.0 * sigma + 0.5);
    std::vector<double> k(2 * r + 1);
    double sum = 0;
    for (int i = -r; i <= r; ++i) { k[i + r] = std::exp(-0.5 * i * i / (sigma * sigma)); sum += k[i + r]; }
    for (double& x : k) x /= sum;
    Plane tmp{in.w, in.h, std::vector<float>(in.v.size())}, out{in.w, in.h, std::vector<float>(in.v.size())};
    for (int y = 0; y < in.h; ++y)
        for (int x = 0; x < in.w; ++x) {
            double a = 0;
            for (int i = -r; i <= r; ++i) a += k[i + r] * in.at(reflect(y + i, in.h), x);
            tmp.v[static_cast<size_t>(y) * in.w + x] = static_cast<float>(a);
        }
    for (int y = 0; y < in.h; ++y)
        for (int x = 0; x < in.w; ++x) {
            double a = 0;
            for (int i = -r; i <= r; ++i) a += k[i + r] * tmp.at(y, reflect(x + i, in.w));
            out.v[static_cast<size_t>(y) * in.w + x] = static_cast<float>(a);
        }
    return out;
}

struct Cand { int o, i, y, x; float mag; };

bool read_token(FILE* f, int& out)
{
    int c = fgetc(f);
    for (;;) {
        while (c == ' ' || c == '\n' || c == '\r' || c == '\t') c = fgetc(f);
        if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
        break;
    }
    if (c < '0' || c > '9') return false;
    long v = 0;
    while (c >= '0' && c <= '9') { v = v * 10 + (c - '0'); if (v > 100000000) return false; c = fgetc(f); }
    out = static_cast<int>(v);
    return true;                                  // the single whitespace behind the token is consumed
}

}  // namespace

bool load_pnm_gray(const std::string& path, Image& img, std::string& err)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    char magic[2] = {0, 0};
    int w = 0, h = 0, maxv = 0;
    bool ok = fread(magic, 1, 2, f) == 2 && magic[0] == 'P' && (magic[1] == '5' || magic[1] == '6') && read_token(f, w) &&
              read_token(f, h) && read_token(f, maxv) && w > 0 && h > 0 && maxv > 0 && maxv <= 255 &&
              static_cast<long long>(w) * h <= 100000000LL;
    if (ok) {
        const int ch = magic[1] == '6' ? 3 : 1;
        std::vector<unsigned char> raw(static_cast<size_t>(w) * h * ch);
        ok = fread(raw.data(), 1, raw.size(), f) == raw.size();
        if (ok) {
            img.w = w; img.h = h;
            img.px.resize(static_cast<size_t>(w) * h);
            if (ch == 1) img.px = raw;
            else
                for (size_t i = 0; i < img.px.size(); ++i)   // PIL "L": (299 R + 587 G + 114 B + 500) / 1000
                    img.px[i] = static_cast<unsigned char>((299 * raw[3 * i] + 587 * raw[3 * i + 1] + 114 * raw[3 * i + 2] + 500) / 1000);
        }
    }
    fclose(f);
    if (!ok) err = path + " is not a binary 8-bit PGM / PPM";
    return ok;
}

Features detect_and_describe(const Image& img, int max_kp, float contrast, float edge_r)
{
    Features out;
    if (img.w < 32 || img.h < 32) return out;
    constexpr int S = 3;
    constexpr double SIGMA0 = 1.6;
    const double kf = std::pow(2.0, 1.0 / S);
    const int n_oct = std::max(1, static_cast<int>(std::log2(static_cast<double>(std::min(img.w, img.h)))) - 4);

    // ---- Gaussian scale space: S + 3 levels per octave, the next octave starts from level S at half resolution
    std::vector<std::vector<Plane>> oct(n_oct);
    Plane base{img.w, img.h, std::vector<float>(img.px.size())};
    for (size_t i = 0; i < img.px.size(); ++i) base.v[i] = static_cast<float>(img.px[i]) / 255.0f;
    base = gaussian(base, std::sqrt(std::max(SIGMA0 * SIGMA0 - 0.25, 0.01)));
    for (int o = 0; o < n_oct; ++o) {
        oct[o].push_back(base);
        for (int i = 1; i < S + 3; ++i) {
            const double sp = SIGMA0 * std::pow(kf, i - 1), st = sp * kf;
            oct[o].push_back(gaussian(oct[o].back(), std::sqrt(st * st - sp * sp)));
        }
        const Plane& src = oct[o][S];
        Plane half{(src.w + 1) / 2, (src.h + 1) / 2, {}};
        half.v.resize(static_cast<size_t>(half.w) * half.h);
        for (int y = 0; y < half.h; ++y)
            for (int x = 0; x < half.w; ++x) half.v[static_cast<size_t>(y) * half.w + x] = src.at(2 * y, 2 * x);
        base = half;
        if (base.w < 20 || base.h < 20) { oct.resize(o + 1); break; }
    }

    // ---- difference-of-Gaussian extrema with contrast and edge tests
    std::vector<Cand> cands;
    for (size_t o = 0; o < oct.size(); ++o) {
        const int w = oct[o][0].w, h = oct[o][0].h;
        std::vector<Plane> dog(S + 2, Plane{w, h, std::vector<float>(static_cast<size_t>(w) * h)});
        for (int i = 0; i < S + 2; ++i)
            for (size_t p = 0; p < dog[i].v.size(); ++p) dog[i].v[p] = oct[o][i + 1].v[p] - oct[o][i].v[p];
        for (int i = 1; i < S + 1; ++i)
            for (int y = 8; y < h - 8; ++y)
                for (int x = 8; x < w - 8; ++x) {
                    const float c = dog[i].at(y, x);
                    if (!(std::fabs(c) > contrast / S)) continue;
                    bool is_max = true, is_min = true;
                    for (int di = -1; di <= 1 && (is_max || is_min); ++di)
                        for (int dy = -1; dy <= 1; ++dy)
                            for (int dx = -1; dx <= 1; ++dx) {
                                const float v = dog[i + di].at(y + dy, x + dx);
                                if (v > c) is_max = false;
                                if (v < c) is_min = false;
                            }
                    if (!is_max && !is_min) continue;
                    const Plane& d = dog[i];
                    const float dxx = d.at(y, x + 1) + d.at(y, x - 1) - 2 * c;
                    const float dyy = d.at(y + 1, x) + d.at(y - 1, x) - 2 * c;
                    const float dxy = (d.at(y + 1, x + 1) - d.at(y + 1, x - 1) - d.at(y - 1, x + 1) + d.at(y - 1, x - 1)) / 4.0f;
                    const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
                    if (det <= 0 || tr * tr * edge_r >= (edge_r + 1) * (edge_r + 1) * det) continue;
                    cands.push_back(Cand{static_cast<int>(o), i, y, x, std::fabs(c)});
                }
    }
    std::stable_sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.mag > b.mag; });
    if (static_cast<int>(cands.size()) > max_kp) cands.resize(max_kp);

    // ---- orientation + 4x4x8 descriptor
    const double PI = 3.14159265358979323846;
    for (const Cand& kp : cands) {
        const Plane& L = oct[kp.o][kp.i];
        const double sig = SIGMA0 * std::pow(kf, kp.i);
        const int rad = static_cast<int>(std::nearbyint(3 * 1.5 * sig));
        const int x = kp.x, y = kp.y;
        if (y - rad < 1 || x - rad < 1 || y + rad >= L.h - 1 || x + rad >= L.w - 1) continue;
        double hist[36] = {0};
        for (int dy = -rad; dy <= rad; ++dy)
            for (int dx = -rad; dx <= rad; ++dx) {
                const float gx = (L.at(y + dy, x + dx + 1) - L.at(y + dy, x + dx - 1)) * 0.5f;
                const float gy = (L.at(y + dy + 1, x + dx) - L.at(y + dy - 1, x + dx)) * 0.5f;
                const float mag = std::sqrt(gx * gx + gy * gy);
                const float ang = std::atan2(gy, gx);
                const double wgt = std::exp(-(dx * dx + dy * dy) / (2 * (1.5 * sig) * (1.5 * sig))) * mag;
                const int bin = static_cast<int>((ang + PI) / (2 * PI) * 36) % 36;
                hist[bin] += wgt;
            }
        double sm[36];
        for (int b = 0; b < 36; ++b) sm[b] = (hist[(b + 35) % 36] + hist[b] + hist[(b + 1) % 36]) / 3.0;
        int best = 0;
        for (int b = 1; b < 36; ++b) if (sm[b] > sm[best]) best = b;
        const double theta = (best + 0.5) / 36 * 2 * PI - PI;

        const double cell = 3.0 * sig;
        const int r2 = static_cast<int>(std::ceil(cell * 2.5 * std::sqrt(2.0))) + 1;
        if (y - r2 < 1 || x - r2 < 1 || y + r2 >= L.h - 1 || x + r2 >= L.w - 1) continue;
        const double c = std::cos(theta), sn = std::sin(theta);
        double desc[4][4][8] = {};
        for (int dy = -r2; dy <= r2; ++dy)
            for (int dx = -r2; dx <= r2; ++dx) {
                const double u = (c * dx + sn * dy) / cell + 1.5;          // cell coordinates in [-0.5, 3.5]
                const double v = (-sn * dx + c * dy) / cell + 1.5;
                if (!(u > -1 && u < 4 && v > -1 && v < 4)) continue;
                const float gx = (L.at(y + dy, x + dx + 1) - L.at(y + dy, x + dx - 1)) * 0.5f;
                const float gy = (L.at(y + dy + 1, x + dx) - L.at(y + dy - 1, x + dx)) * 0.5f;
                const double mag = std::sqrt(gx * gx + gy * gy) * std::exp(-((u - 1.5) * (u - 1.5) + (v - 1.5) * (v - 1.5)) / (2 * 2.0 * 2.0));
                double a = std::fmod(std::atan2(gy, gx) - theta, 2 * PI);
                if (a < 0) a += 2 * PI;
                const double ob = a / (2 * PI) * 8;
                const int u0 = static_cast<int>(std::floor(u)), v0 = static_cast<int>(std::floor(v)), o0 = static_cast<int>(std::floor(ob));
                const double du = u - u0, dv = v - v0, dob = ob - o0;
                for (int a1 = 0; a1 < 2; ++a1)
                    for (int b1 = 0; b1 < 2; ++b1)
                        for (int e1 = 0; e1 < 2; ++e1) {
                            const int vi = v0 + a1, ui = u0 + b1, oi = (o0 + e1) % 8;
                            if (vi < 0 || vi >= 4 || ui < 0 || ui >= 4) continue;
                            desc[vi][ui][oi] += mag * (a1 ? dv : 1 - dv) * (b1 ? du : 1 - du) * (e1 ? dob : 1 - dob);
                        }
            }
        const double* d = &desc[0][0][0];
        double nrm = 0;
        for (int i = 0; i < 128; ++i) nrm += d[i] * d[i];
        nrm = std::sqrt(nrm);
        if (nrm < 1e-9) continue;
        double cl[128], n2 = 0;
        for (int i = 0; i < 128; ++i) { cl[i] = std::min(d[i] / nrm, 0.2); n2 += cl[i] * cl[i]; }
        n2 = std::sqrt(n2);
        for (int i = 0; i < 128; ++i) {
            double q = std::nearbyint(cl[i] / n2 * 512);
            q = q < 0 ? 0 : (q > 255 ? 255 : q);
            out.desc.push_back(static_cast<float>(q));
        }
        const float scale = static_cast<float>(1 << kp.o);
        out.kp_xy.push_back(x * scale);
        out.kp_xy.push_back(y * scale);
        ++out.n;
    }
    return out;
}

}  // namespace pm_feat
