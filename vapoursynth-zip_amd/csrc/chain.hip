// vszip_chain_run: several pixel-filter stages over one table of resident planes in ONE call — what a host
// does for a script that chains vszip filters (clip.vszip.Bilateral().vszip.BoxBlur()...): upload once, run every
// stage on the device, download once. Every entry point of this library already works on device pointers, so
// a chain is a sequence of calls on one context; this one adds the intermediate planes (a grow-only buffer of the
// context, two planes per table entry, ping-pong) and the bookkeeping of which stage wrote which plane, so that
// the host side needs no device allocations of its own. Stages follow the reference's plane semantics: a stage
// filters the planes it is told to and passes the others through (newVideoFrame2's plane copy,
// src/vapoursynth/boxblur.zig:38).
#include <vector>

#include "common.hpp"

static int ensure_chain_buf(vszip_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->chain_bytes) return VSZIP_OK;
    if (ctx->chain_buf) {
        VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->chain_buf);
        ctx->chain_buf = nullptr;
        ctx->chain_bytes = 0;
    }
    const size_t want = bytes + (bytes >> 3) + 4096;
    if (vszip_hip_malloc(ctx, &ctx->chain_buf, want) != hipSuccess) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "chain buffer allocation of %zu bytes failed", want);
    ctx->chain_bytes = want;
    return VSZIP_OK;
}

void vszip_chain_release(vszip_ctx *ctx) {
    if (ctx->chain_buf) (void)hipFree(ctx->chain_buf);
    ctx->chain_buf = nullptr;
    ctx->chain_bytes = 0;
}

VSZIP_EXPORT int vszip_chain_run(vszip_ctx *ctx, int dtype, const vszip_chain_stage *stages, int nstages, const vszip_plane *planes, const int *plane_slot, int nplanes) {
    if (!ctx || !stages || !planes || !plane_slot || nstages <= 0 || nplanes <= 0) return VSZIP_ERR_ARG;
    const int bps = vszip_dtype_size(dtype);
    if (bps <= 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "chain: sample type %d", dtype);
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < nplanes; ++i)
        if (!planes[i].src || !planes[i].dst || planes[i].w <= 0 || planes[i].h <= 0 || plane_slot[i] < 0 || plane_slot[i] > 2)
            return vszip_set_error(ctx, VSZIP_ERR_ARG, "chain: bad plane %d", i);
    // how many stages write each plane, and the two intermediate planes of every entry (dense rows, 256-byte pitch)
    std::vector<int> writers(nplanes, 0);
    std::vector<size_t> off(nplanes + 1, 0), pitch(nplanes);
    for (int i = 0; i < nplanes; ++i) {
        for (int s = 0; s < nstages; ++s) writers[i] += stages[s].process[plane_slot[i]] ? 1 : 0;
        pitch[i] = ((size_t)planes[i].w * bps + 255) & ~(size_t)255;
        off[i + 1] = off[i] + (writers[i] > 1 ? 2 * pitch[i] * planes[i].h : 0);
    }
    int rc = ensure_chain_buf(ctx, off[nplanes]);
    if (rc != VSZIP_OK) return rc;
    char *buf = static_cast<char *>(ctx->chain_buf);
    std::vector<const void *> cur(nplanes);
    std::vector<ptrdiff_t> cur_stride(nplanes);
    std::vector<int> written(nplanes, 0);
    for (int i = 0; i < nplanes; ++i) {
        cur[i] = planes[i].src;
        cur_stride[i] = planes[i].src_stride;
    }
    std::vector<vszip_plane> tab;
    std::vector<int> idx;
    for (int s = 0; s < nstages; ++s) {
        const vszip_chain_stage &st = stages[s];
        tab.clear();
        idx.clear();
        for (int i = 0; i < nplanes; ++i) {
            if (!st.process[plane_slot[i]]) continue;
            vszip_plane p = planes[i];
            p.src = cur[i];
            p.src_stride = cur_stride[i];
            p.ref = nullptr;
            p.ref_stride = 0;
            if (written[i] + 1 == writers[i]) {  // the last stage that writes this plane writes the caller's output
                p.dst = planes[i].dst;
                p.dst_stride = planes[i].dst_stride;
            } else {
                p.dst = buf + off[i] + (size_t)(written[i] & 1) * pitch[i] * planes[i].h;
                p.dst_stride = (ptrdiff_t)(pitch[i] / bps);
            }
            tab.push_back(p);
            idx.push_back(i);
        }
        if (tab.empty()) continue;
        const int n = (int)tab.size();
        switch (st.kind) {
            case VSZIP_STAGE_BOXBLUR:
                rc = vszip_boxblur(ctx, dtype, tab.data(), n, st.hradius, st.hpasses, st.vradius, st.vpasses);
                break;
            case VSZIP_STAGE_BILATERAL: {
                std::vector<const vszip_bilateral_cfg *> cfgs(n);
                for (int k = 0; k < n; ++k) {
                    cfgs[k] = st.bilateral[plane_slot[idx[k]]];
                    if (!cfgs[k]) return vszip_set_error(ctx, VSZIP_ERR_ARG, "chain: stage %d has no Bilateral configuration for plane slot %d", s, plane_slot[idx[k]]);
                }
                rc = vszip_bilateral(ctx, dtype, tab.data(), cfgs.data(), n, st.peak);
                break;
            }
            case VSZIP_STAGE_LIMITER: {
                std::vector<double> lo(n), hi(n);
                for (int k = 0; k < n; ++k) {
                    lo[k] = st.lo[plane_slot[idx[k]]];
                    hi[k] = st.hi[plane_slot[idx[k]]];
                }
                rc = vszip_limiter(ctx, dtype, tab.data(), n, lo.data(), hi.data());
                break;
            }
            default:
                return vszip_set_error(ctx, VSZIP_ERR_ARG, "chain: stage kind %d", st.kind);
        }
        if (rc != VSZIP_OK) return rc;
        for (int k = 0; k < n; ++k) {
            const int i = idx[k];
            cur[i] = tab[k].dst;
            cur_stride[i] = tab[k].dst_stride;
            ++written[i];
        }
    }
    // planes no stage filters pass through
    for (int i = 0; i < nplanes; ++i)
        if (writers[i] == 0 && planes[i].dst != planes[i].src) {
            rc = vszip_copy_d2d_2d(ctx, planes[i].dst, (size_t)planes[i].dst_stride * bps, planes[i].src, (size_t)planes[i].src_stride * bps, (size_t)planes[i].w * bps, (size_t)planes[i].h);
            if (rc != VSZIP_OK) return rc;
        }
    return VSZIP_OK;
}
