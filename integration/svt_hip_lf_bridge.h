/*
 * svt_hip_lf_bridge.h — reference-side glue for SURVEY 8(f) rank 1, the three in-loop filter process loops:
 *   dlf_kernel   (Source/Lib/Encoder/Codec/EbDlfProcess.c:175-216)  svt_av1_loop_filter_frame          -> svt_hip_dlf_picture()
 *   cdef_kernel  (EbCdefProcess.c:510-534)  cdef_seg_search[16bit] per segment + svt_av1_cdef_frame      -> svt_hip_cdef_search_picture(), svt_hip_cdef_apply_picture()
 *   rest_kernel  (EbRestProcess.c:527)      restoration_seg_search: search_sgrproj_seg per unit            -> svt_hip_sgr_search_picture()
 *   rest_kernel  (EbRestProcess.c:548)      svt_av1_loop_restoration_filter_frame                        -> svt_hip_rest_apply_picture()
 * Each replaces a per-SB / per-segment loop by one batched call per picture and leaves the reference's own objects (recon picture,
 * pcs->mse_seg, cm->rst_info) exactly as the C loops leave them, so finish_cdef_search, the restoration search's host logic and the
 * FIFOs are untouched.  8-bit and 16-bit pictures.
 *
 * Compiled INTO libSvtAv1Enc (it includes the reference's headers); not part of libsvtav1_hip.so.
 * tests/test_integration_compiles.py syntax-checks it against /root/reference when that tree exists.
 */
#ifndef SVT_HIP_LF_BRIDGE_H
#define SVT_HIP_LF_BRIDGE_H

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbPictureBufferDesc.h"
#include "svt_hip.h"

/* Device-resident copies of one picture's planes (pixel (0,0) of each plane; strides in samples), allocated once per encoder instance
 * for the largest picture and reused: recon (deblocked in place), cdef output, restoration output, source. */
typedef struct SvtHipLfPicture {
    int   pix_bytes, bd, w, h;           /* luma size, multiples of 8 */
    void *d_recon[3], *d_cdef[3], *d_rest[3], *d_src[3];
    int   stride[3];                     /* recon / cdef / rest share one stride per plane; the planes carry a 3-sample border */
    int   src_stride[3];
    /* CDEF */
    uint8_t  *d_skip8, *h_skip8;         /* [h/8][w/8] is_8x8_block_skip */
    uint64_t *d_mse;                     /* [2][nfb][64] */
    uint8_t  *d_dir;  int32_t *d_var;    /* [nfb * 64] */
    uint8_t  *d_y_strength, *d_uv_strength;
    /* deblocking */
    SvtHipDlfModeInfo *h_mi;             /* [mi_rows][mi_cols] */
    uint16_t *h_edges[3][2], *d_edges[3][2];
    int       units_w[3], units_h[3];
    /* restoration */
    uint8_t *d_unit_ep[3]; int32_t *d_unit_xqd[3]; int16_t *d_unit_wiener[3];
} SvtHipLfPicture;

EbErrorType svt_hip_lf_picture_ctor(SvtHipCtx *hip, SvtHipLfPicture *p, int w, int h, int is_16bit, int bd);
void        svt_hip_lf_picture_dctor(SvtHipCtx *hip, SvtHipLfPicture *p);

/* recon <-> device (plane by plane, the picture's own strides) */
EbErrorType svt_hip_lf_upload(SvtHipCtx *hip, SvtHipLfPicture *p, const EbPictureBufferDesc *pic, void *const d_dst[3]);
EbErrorType svt_hip_lf_download(SvtHipCtx *hip, const SvtHipLfPicture *p, void *const d_src[3], EbPictureBufferDesc *pic);

/* dlf_kernel: after svt_av1_pick_filter_level has set frm_hdr->loop_filter_params, in place of svt_av1_loop_filter_frame(recon, pcs, 0, 3).
 * p->d_recon must hold the reconstruction; it holds the deblocked picture afterwards (kept resident for CDEF and restoration). */
EbErrorType svt_hip_dlf_picture(SvtHipCtx *hip, SvtHipLfPicture *p, PictureControlSet *pcs);

/* cdef_kernel: in place of every cdef_seg_search / cdef_seg_search16bit call of the picture; fills pcs->mse_seg[2][fb][64] */
EbErrorType svt_hip_cdef_search_picture(SvtHipCtx *hip, SvtHipLfPicture *p, PictureControlSet *pcs);
/* after finish_cdef_search: in place of svt_av1_cdef_frame / av1_cdef_frame16bit; p->d_cdef receives the filtered picture */
EbErrorType svt_hip_cdef_apply_picture(SvtHipCtx *hip, SvtHipLfPicture *p, PictureControlSet *pcs);

/* rest_kernel: in place of svt_av1_loop_restoration_save_boundary_lines (x2) + svt_av1_loop_restoration_filter_frame; the stripe context
 * rows come straight from the resident deblocked picture (p->d_recon).  p->d_rest receives the restored picture. */
EbErrorType svt_hip_rest_apply_picture(SvtHipCtx *hip, SvtHipLfPicture *p, PictureControlSet *pcs);

/* rest_kernel, search half: in place of the search_sgrproj_seg calls of restoration_seg_search (every unit of the three planes): fills
 * pcs->parent_pcs_ptr->rusi_picture[plane][unit].sgrproj / .sse[RESTORE_SGRPROJ] and cm->sg_frame_ep_cnt for search_sgrproj_finish / rest_finish_search. */
EbErrorType svt_hip_sgr_search_picture(SvtHipCtx *hip, SvtHipLfPicture *p, PictureControlSet *pcs);

#endif
