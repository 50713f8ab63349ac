// md_pre.hip — mode decision, picture-level precompute ("MD candidate batching").
//
// What the reference does, one block and one candidate at a time (EbProductCodingLoop.c:907 fast_loop_core, called by md_stage_0 :1461 for every candidate
// of every block of every superblock): predict the candidate (EbEncInterPrediction.c:6178 inter_pu_prediction_av1 -> :4040 av1_inter_prediction ->
// svt_inter_predictor) and measure its luma distortion against the source (svt_nxm_sad_kernel_sub_sampled, :953 — despite its name the plain SAD of all rows).
// In the first partitioning pass at presets above M4 the open-loop ME vectors go into stage 0 unrefined (EbEncDecProcess.c:3050-3093: md_sq / nsq / pme /
// sub-pel search levels 0), i.e. every ME candidate is a FULL-PEL, single-reference translation whose prediction is a copy of the reference block
// (svt_av1_convolve_2d_copy_sr).  None of that depends on a neighbouring block: the vectors are the open-loop ME's (known before the picture's mode decision
// starts), the reference pictures are complete, the source is the input.  So ONE launch per picture computes the distortion of every (superblock, square PU,
// reference picture) triple; the patched fast_loop_core reads it from a table (integration/svt_hip_md_bridge.c, hook "md_pre").
//
// Layout: one 256-thread workgroup per (superblock, reference picture); its four waves take the PUs of the list round-robin.  A wave covers a PU with one
// dword (four samples) per lane and iteration — lanes (w / 4) per row, 64 / (w / 4) rows per iteration — on v_sad_u8; neither plane has to be aligned
// (two aligned dwords + v_alignbyte).  The superblock's 4 KB of source and the <= 85 reference blocks around it stay in the L1 / L2 of the workgroup's CU.
// HBM-bound in principle (algorithmic bytes per (SB, reference): 4096 source + ~4 x 4096 reference samples + 85 x 8 table bytes), launch-bound in practice:
// a 1080p picture with 7 references is 3 570 workgroups of ~30 iterations.
#include "svt_hip_internal.h"

namespace {

struct RefPlanes { SvtHipMdRefPlane r[SVT_HIP_MD_MAX_REFS]; };
struct PuList { uint8_t x[SVT_HIP_MD_MAX_PUS], y[SVT_HIP_MD_MAX_PUS], w[SVT_HIP_MD_MAX_PUS], h[SVT_HIP_MD_MAX_PUS]; };

// four consecutive samples starting at any byte address
__device__ __forceinline__ uint32_t load4_any(const uint8_t* p) {
    const uintptr_t a = (uintptr_t)p;
    const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
    const uint32_t lo = q[0], hi = q[1];
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(a & 3));
}

__global__ void __launch_bounds__(256)
md_fullpel_sad_kernel(const uint8_t* __restrict__ src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_pus, int n_refs, RefPlanes refs, PuList pus,
                      const uint32_t* __restrict__ mv, uint32_t* __restrict__ sad) {
    const int sb = blockIdx.x, r = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sb_x = (sb % sb_cols) * 64, sb_y = (sb / sb_cols) * 64;
    const SvtHipMdRefPlane ref = refs.r[r];
    for (int pu = wave; pu < n_pus; pu += 4) {
        const size_t slot = ((size_t)sb * n_pus + pu) * n_refs + r;
        const int w = pus.w[pu], h = pus.h[pu], x = sb_x + pus.x[pu], y = sb_y + pus.y[pu];
        const uint32_t m = mv[slot];
        const int mx = (int16_t)(m & 0xffff), my = (int16_t)(m >> 16);
        const int rx = x + mx, ry = y + my;
        // no vector, a PU that leaves the picture, a reference block that leaves the plane's allocation (the last four samples of a row pair are read as whole
        // dwords): the slot says "not computed"
        const bool ok = mx != SVT_HIP_MD_NO_MV && x + w <= pic_w && y + h <= pic_h && rx >= ref.x_min && ry >= ref.y_min && rx + w + 4 <= ref.x_max && ry + h <= ref.y_max;
        if (!ok) {   // wave-uniform
            if (lane == 0) sad[slot] = 0xffffffffu;
            continue;
        }
        const int n4 = w >> 2, rows = 64 / n4, row = lane / n4, c4 = (lane - row * n4) << 2;
        const uint8_t* ps = src + (ptrdiff_t)y * src_stride + x + c4;
        const uint8_t* pr = ref.d_plane + (ptrdiff_t)ry * ref.stride + rx + c4;
        uint32_t s = 0;
        for (int r0 = 0; r0 < h; r0 += rows) {
            const int yy = r0 + row;
            if (yy < h) s = __builtin_amdgcn_sad_u8(load4_any(ps + (ptrdiff_t)yy * src_stride), load4_any(pr + (ptrdiff_t)yy * ref.stride), s);
        }
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) s += (uint32_t)__shfl_xor((int)s, k, 64);
        if (lane == 0) sad[slot] = s;
    }
}

}   // namespace

extern "C" int svt_hip_launch_md_fullpel_sad(hipStream_t st, const uint8_t* src, int src_stride, int pic_w, int pic_h, int sb_cols, int n_sb, int n_pus,
                                             const SvtHipMdPu* pus, int n_refs, const SvtHipMdRefPlane* refs, const uint32_t* mv, uint32_t* sad) {
    if (n_sb <= 0 || n_refs <= 0 || n_pus <= 0) return 0;
    RefPlanes rp;
    PuList    pl;
    for (int i = 0; i < SVT_HIP_MD_MAX_REFS; i++) rp.r[i] = refs[i < n_refs ? i : 0];
    for (int i = 0; i < SVT_HIP_MD_MAX_PUS; i++) {
        const SvtHipMdPu p = pus[i < n_pus ? i : 0];
        pl.x[i] = p.x; pl.y[i] = p.y; pl.w[i] = p.w; pl.h[i] = p.h;
    }
    hipLaunchKernelGGL(md_fullpel_sad_kernel, dim3(n_sb, n_refs), dim3(256), 0, st, src, src_stride, pic_w, pic_h, sb_cols, n_pus, n_refs, rp, pl, mv, sad);
    return (int)hipGetLastError();
}

SVT_HIP_TU_PROBE(md_pre)
