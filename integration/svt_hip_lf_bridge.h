/*
 * svt_hip_lf_bridge.h — reference-side glue for SURVEY 8(f) rank 1, the three in-loop filter process loops (the hook entry points
 * themselves are declared in svt_hip_hooks.h):
 *   dlf_kernel   (Source/Lib/Encoder/Codec/EbDlfProcess.c:175-216)  svt_av1_pick_filter_level / svt_av1_loop_filter_frame
 *   cdef_kernel  (EbCdefProcess.c:510-534)  cdef_seg_search[16bit] per segment, svt_av1_cdef_frame / av1_cdef_frame16bit
 *   rest_kernel  (EbRestProcess.c:527-548)  restoration_seg_search (search_sgrproj_seg, svt_av1_compute_stats per unit),
 *                                           svt_av1_loop_restoration_filter_frame
 * Each hook replaces a per-SB / per-segment / per-unit loop by batched calls over the whole picture and leaves the reference's own
 * objects (recon picture, frm_hdr->loop_filter_params, pcs->mse_seg, rusi_picture, cm->rst_info) exactly as the C loops leave them, so
 * finish_cdef_search, rest_finish_search, the FIFOs and the entropy coder are untouched.
 *
 * One SvtHipLfPicture per picture in flight keeps the planes on the device between the stages: source, reconstruction -> deblocked in
 * place (kept: CDEF input, and the stripe context rows of the restoration filters), CDEF output, restoration output.  With a subset of the hooks
 * active every hook downloads its result into the reference's recon picture as well (the host state is always what the C path would have produced,
 * and a stage whose hook is off simply finds its input on the host as usual); with ALL of them active the picture is "deferred": it comes back
 * once, when it leaves the filter stages, and a hook that fails on the way first brings the host up to date (svt_hip_lf_bridge.c, lf_recover).
 *
 * Compiled INTO libSvtAv1Enc (it includes the reference's headers); not part of libsvtav1_hip.so.
 */
#ifndef SVT_HIP_LF_BRIDGE_H
#define SVT_HIP_LF_BRIDGE_H

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbPictureBufferDesc.h"
#include "svt_hip.h"

typedef struct SvtHipLfPicture {
    int   pix_bytes, bd, w, h;           /* coded luma size, multiples of 8 */
    int   cw, ch;                        /* luma size without the padding of a source that is not a multiple of 8 (w - max_input_pad_right, h - _bottom):
                                          * what the restoration stages work on (link_eb_to_aom_buffer_desc's crop size) */
    int   sb_size;                       /* 64 / 128: superblock size of the sequence (deblocking range of the last superblock row / column) */
    void *d_recon[3], *d_cdef[3], *d_rest[3], *d_src[3];
    void *d_dbl[3];                      /* destination of the one-launch (out-of-place) deblocking: swapped with d_recon afterwards, so d_recon stays "the deblocked picture" */
    int   stride[3];                     /* recon / cdef / rest share one stride per plane; the planes carry a 3-sample border */
    int   src_stride[3];
    void *src[3];                        /* the source planes the stages read: d_src (uploaded, src_stride) or the picture's resident copy in place (svt_hip_resident.h) */
    int   src_st[3];
    /* CDEF */
    uint8_t  *d_skip8, *h_skip8;         /* [h/8][w/8] is_8x8_block_skip */
    uint64_t *d_mse, *h_mse;             /* [2][nfb][64] */
    uint8_t  *d_dir;  int32_t *d_var;    /* [nfb * 64] */
    uint8_t  *d_y_strength, *d_uv_strength;
    /* deblocking */
    SvtHipDlfModeInfo *h_mi, *d_mi;      /* [mi_rows][mi_cols]; the device copy serves the level search and the filter of one picture */
    int       h_mi_pinned;
    uint16_t *h_mi_until;                /* [mi_cols] scratch of the grid's fill pass */
    uint8_t  *h_skip4;                   /* [mi_rows][mi_cols] the skip flag of every 4 x 4 unit, written by the same pass */
    uint16_t *h_edges[3][2], *d_edges[3][2];
    int       units_w[3], units_h[3];
    uint64_t *d_sse;
    /* restoration */
    int      max_units[3];
    uint8_t *d_unit_ep[3]; int32_t *d_unit_xqd[3]; int16_t *d_unit_wiener[3];
    int64_t *h_wiener_M[3], *h_wiener_H[3];  /* picture-level Wiener statistics, [unit][win^2] / [unit][win^4] */
    int      wiener_win[3];
} SvtHipLfPicture;

EbErrorType svt_hip_lf_picture_ctor(SvtHipCtx *hip, SvtHipLfPicture *p, int w, int h, int is_16bit, int bd);
void        svt_hip_lf_picture_dctor(SvtHipCtx *hip, SvtHipLfPicture *p);

#endif
