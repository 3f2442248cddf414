// TEST INFRASTRUCTURE ONLY (see oracle_common.h). CPU restatement of vszip.AdaptiveBinarize.
//
// Follows (vszip v19.0.0): src/vapoursynth/adaptive_binarize.zig:26-73 (getFrame: dst = 255 where
// clip2 - clip >= c, else 0, in i16, every plane) and :96-99 (c clamped to [-256, 256]).
// Parity unpinned by goldens: the reference's cases build clip2 with the VapourSynth core's
// std.BoxBlur, which is outside the reference repo; pinned by inspection and the binary-output test.
#include "oracle_common.h"

VSZO_API int vszo_adaptive_binarize(const uint8_t* src, const uint8_t* src2, uint8_t* dst, ptrdiff_t s1, ptrdiff_t s2, ptrdiff_t ds, int w, int h, int c) {
    const int16_t cc = (int16_t)(c < -256 ? -256 : (c > 256 ? 256 : c));
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) dst[y * ds + x] = ((int16_t)src2[y * s2 + x] - (int16_t)src[y * s1 + x] >= cc) ? 255 : 0;
    return 0;
}
