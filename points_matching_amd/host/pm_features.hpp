// pm_features.hpp — build-owned keypoint detector + 128-D descriptor for the image-pair-in surface of pm_cli
// (SURVEY.md 8f-2).  Slot in the reference: main.cpp:22-26 (SurfFeatureDetector(8000).detect) and main.cpp:36-40
// (SurfDescriptorExtractor().compute); BASELINE config C1 names SIFT-128, and OpenCV's nonfree module is neither in
// /root/reference nor in this image, so this is NOT a SURF/SIFT clone: it follows Lowe's published scheme — Gaussian
// scale space (3 scales per octave, sigma0 1.6), difference-of-Gaussian extrema, contrast + edge rejection, dominant
// gradient orientation, 4x4x8 gradient histogram with trilinear voting, normalise -> clip 0.2 -> renormalise -> x512
// -> saturate to 0..255 — and produces what the matcher's exact f16 route expects of SIFT: u8-valued float rows.
// Host C++ (the reference's front-end is host code too; the hot path starts at the descriptors).  It is the C++
// twin of tools/sift_numpy.py, which made the C1 fixture; the two agree up to filter-tap rounding (tests compare
// keypoint sets and the pipeline's result, not bits).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pm_feat {

struct Image {               // 8-bit grey, row-major
    int w = 0, h = 0;
    std::vector<unsigned char> px;
};

struct Features {
    std::vector<float> kp_xy;       // n x 2 pixel coordinates (x, y) in the input image
    std::vector<float> desc;        // n x 128, integer-valued 0..255
    int n = 0;
};

// Binary PGM (P5, maxval <= 255) or binary PPM (P6; converted with the ITU-R 601 luma weights PIL's "L" uses).
bool load_pnm_gray(const std::string& path, Image& img, std::string& err);

// max_kp strongest DoG extrema (by |response|) that survive the tests; deterministic.
Features detect_and_describe(const Image& img, int max_kp = 4000, float contrast = 0.03f, float edge_r = 10.0f);

}  // namespace pm_feat
