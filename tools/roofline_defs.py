"""ONE definition of the roofline arithmetic, shared by bench.py (live stage times from HIP events) and tools/summarize_profiles.py (rocprofv3 durations of
the committed profile): every number of the bench line's `roofline` object can be recomputed from profiles/<round>/{pmc_traffic.json,kernel_stats.csv}
with the functions below.  DESIGN.md section 6 and docs/design/measurement.md quote it.

Peaks (/opt/skills/guides/MI355X_MICROARCH.md): HBM3E 8.0 TB/s; integer VALU 256 CUs x 128 lanes x 2.4 GHz = 78.6 T lane-operations/s (one lane-operation =
one lane of one VALU instruction, whatever it computes).

Per stage and superblock (8-bit 4:2:0, 6144 samples):
  algorithmic bytes  SURVEY.md 8(d): every input byte read once, every output byte written once
  useful operations  the lane-operations an ideal kernel would still have to issue (searches only); ONE derivation each, spelled out as code below (USEFUL_*):
      me_fullpel_85pu      4096 samples x 4096 candidates absolute differences, 4 per lane-operation (v_sad_u8)
      cdef_search          6144 samples x 64 strength pairs x 2 (combine + round + clamp + squared error on packed 16-bit pairs)
      sgr_units_search     6144 samples x SGR_OPS_PER_SAMPLE (the 23 distinct box-filter passes of the 16 sets + projection sums + the reference's own walk, see there)
      cdef_strength_select 75 steps x 4096 pairs x 3 (add, min, accumulate)
  issued operations  SQ_INSTS_VALU x 64 of the stage's kernels (PMC pass of the profile), per frame
  traffic            FETCH_SIZE x read factor + WRITE_SIZE x write factor of the stage's kernels per frame; the factors are MEASURED on this chip by
                     tools/calibrate_counters.sh (kernels that move a known byte count at 1 / 2 / 4 / 8 / 16 bytes per lane): profiles/<round>/counter_calibration.json --
                     FETCH_SIZE reports half of the bytes read at EVERY access width (2.000), WRITE_SIZE is exact (1.000; 0.984 for byte stores)

The bench line's `roofline.frac` is the USEFUL fraction (algorithmic work / stage time / peak) of the bound that binds the dominant stage; `issued_frac` (what the ISA issued)
stands beside it -- it rewards waste and is not the roofline figure.
"""
HBM_PEAK_BPS = 8.0e12
VALU_PEAK_LANE_OPS = 256 * 128 * 2.4e9
# What the chip actually issues, by WALL CLOCK (tools/ubench/valu_peak.hip, profiles/r06/valu_peak.txt: 256 / 512 workgroups of 1024 threads, 16 independent instances of
# one instruction per wave): v_mad_u32_u24, v_and_or_b32, v_perm_b32, v_pk_add_u16, v_dot2_i32_i16, v_sad_u8 / u16 all run at 36-37.5 T lane-operations per second
# (~61 lanes per clock and compute unit: a wave64 instruction occupies its SIMD for four cycles), v_add_u32 at 55 T, v_fma_f32 at 40 T, v_qsad_pk_u16_u8 at 9.6 T.
# The 78.6 T above is the guide's FP32 vector peak, which counts packed / dual-issued FP32; the integer and packed-16-bit instructions these kernels are made of do not
# get it.  (tools/valu_rate.cpp of round 2 divided s_memtime deltas by instruction counts and reported 23 lanes per clock and SIMD for the same instructions; by the
# wall clock it is 15: the counter does not tick once per shader cycle.)  `frac` stays the fraction of the guide's peak; the *_of_measured_peak fields use this one.
VALU_MEASURED_LANE_OPS = 37.3e12

ALG_BYTES_PER_SB = {
    "me_fullpel_85pu": 8872,                 # 4096 src + 4096 ref (amortised) + 85*8 out
    "fwd_txfm_quant": 61440 + 128,           # (src+pred 2*6144) + qcoeff+dqcoeff 2*4*6144 + eob   (luma+chroma)
    "inv_txfm_recon": 36864,                 # dqcoeff 4*6144 + pred 6144 + recon 6144
    "fwd_quant_inv_recon": 2 * 6144 + 4 * 6144 + 6144 + 128,   # fused: src + pred in, levels + recon out (the dequantised coefficients stay in registers)
    "deblock": 6144 + 6144 + 2560,           # SURVEY 8(d) fused figure: both directions in one out-of-place launch (planes R once + W once + edge descriptors); 27 136 for the two in-place passes
    "cdef_search": 13312,                    # recon 6144 + source 6144 R + 2*64*8 W
    "cdef_strength_select": 2 * 64 * 8 + 2,  # the two distortion rows of the filter block R (once, if they stayed on chip over the 75 steps) + its strength pair W
    "cdef_apply": 12288,                     # 6144 R + 6144 W (the kernel writes every sample: no initialising copy)
    "pyramids": 5376 + 4351,                 # decimation 4096 R + 1024 + 256 W ; variance pyramid 4096 R + 85*3 W
    "hme_l0_l1_l2": 256 + 1024 + 4096 + 3 * 12,   # source blocks of the three levels + results (windows are cache-resident)
    "subpel_convolve": 12560,                # 16 blocks x (16+7)^2 R + 4096 W (luma)
    "sgr_units_search": 12288 + 640,         # dgd 6144 + source 6144 R + results: the minimum if everything in between stayed on chip
    "sgr_apply": 12288,                      # 6144 R + 6144 W
}
# ---- restoration unit search: ideal lane-operations per SAMPLE for all 16 parameter sets (eb_sgr_params, Common/Codec/EbRestoration.c:136-153).  The one derivation
# (DESIGN.md section 4 quotes this code): box filters INCLUDED, as SURVEY 8(d) defines the class.
#   filters   sets 0-9 have both radii, 10-13 r = 1 only, 14-15 r = 2 only; 11 / 12 / 13 use the r = 1 parameters of 2 / 5 / 8 -> 11 distinct r = 1 passes + 12 r = 2 passes
#             = the "23 box filters".  Per pass and sample: A/B of a window position (z = p s >> 20, table, B' = t m >> 12, pack: 6; r = 1 has a position per sample, r = 2 one
#             per two samples -- only odd rows carry positions), the weighted neighbourhood on the packed A|B words (r = 1: 9, r = 2: 7 on average over even / odd rows), apply
#             (a x + b, rounding shift: 2).  Window statistics (sums, sums of squares) are shared by all sets and not counted.
#   sums      the five projection products of a two-filter set, two of a one-filter set
#   walk      finer_search_pixel_proj_error (Encoder/Codec/EbRestorationPick.c:353-440) evaluates at least 1 + 4 + 4 points of a two-filter set (start, both directions of both
#             taps at step 2 and at step 1); a point costs 2 lane-operations per sample (a dot product for the weighted sum and its rounding, half a pack, half a squaring
#             dot product).  One-filter sets are evaluated on a histogram: one operation per sample, once.
SGR_FILTER_R1 = 6.0 + 9.0 + 2.0
SGR_FILTER_R2 = 3.0 + 7.0 + 2.0
SGR_OPS_PER_SAMPLE = (11 * SGR_FILTER_R1 + 12 * SGR_FILTER_R2) + (10 * 5 + 6 * 2) + (10 * 9 * 2.0 + 6 * 1.0)   # 331 + 62 + 186 = 579 (36.2 per set)
USEFUL_LANE_OPS_PER_SB = {
    "me_fullpel_85pu": 4096 * 4096 / 4.0,
    "cdef_search": 6144 * 64 * 2.0,
    "sgr_units_search": 6144 * SGR_OPS_PER_SAMPLE,
    "cdef_strength_select": 75 * 4096 * 3.0,
}


def counter_calibration(root=None):
    """(read factor, write factor, source) for FETCH_SIZE / WRITE_SIZE: the latest profiles/<round>/counter_calibration.json (tools/calibrate_counters.sh), else the values
    measured in round 6.  The factors are the same at every access width on gfx950, so one pair serves every kernel."""
    import glob
    import json
    import os
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "*", "counter_calibration.json")))
    if files:
        try:
            c = json.load(open(files[-1]))
            r = [v for v in c["read"].values()]; w = [v for k, v in c["write"].items() if k != "1"]
            return sum(r) / len(r), sum(w) / len(w), os.path.relpath(files[-1], root)
        except (OSError, ValueError, KeyError, ZeroDivisionError):
            pass
    return 2.0, 1.0, "defaults (round-6 measurement)"
# kernel-name prefixes (as tools/summarize_profiles.py shortens them) of each stage
STAGE_KERNELS = {
    "pyramids": ("downsample_kernel", "variance_pyramid_kernel"), "hme_l0_l1_l2": ("sad_loop_kernel",), "me_fullpel_85pu": ("me_fullpel_85pu_kernel", "me_fullpel_narrow_kernel"),
    "subpel_convolve": ("subpel_predict_kernel", "subpel_jobs_from_me_kernel"), "fwd_txfm_quant": ("fwd_txfm_quant_multi_kernel",), "inv_txfm_recon": ("inv_txfm_add_multi_kernel",),
    "fwd_quant_inv_recon": ("enc_txfm_multi_kernel",), "deblock": ("deblock_fused_kernel", "deblock_frame_pass_kernel"), "cdef_search": ("cdef_search_luma_kernel", "cdef_search_chroma_kernel"),
    "cdef_strength_select": ("joint_init_kernel", "joint_partial_kernel", "joint_reduce_kernel", "joint_transpose_kernel", "joint_resident_kernel", "cdef_finish_kernel"), "cdef_apply": ("cdef_apply_kernel",),
    "sgr_units_search": ("sgr_search8_kernel", "sgr_walk_resident_kernel", "sgr_walk_kernel", "generate_padding_kernel"), "sgr_apply": ("lr_apply8_kernel",),
}


def stage_counters(pmc, stage, frames):
    """Per FRAME, from a pmc_traffic.json dict: (kernel microseconds, CORRECTED traffic bytes = FETCH_SIZE x read factor + WRITE_SIZE x write factor, SQ_INSTS_VALU,
    SQ_ACTIVE_INST_VALU) of the stage's kernels."""
    rf, wf, _ = counter_calibration()
    us = traffic = insts = active = 0.0
    for k, e in pmc.items():
        if not any(k.startswith(p) for p in STAGE_KERNELS.get(stage, ())):
            continue
        n = e["launches"] / float(frames)
        us += e["avg_us"] * n
        traffic += (rf * e.get("fetch_bytes_per_launch", 0.0) + wf * e.get("write_bytes_per_launch", 0.0)) * n
        insts += e.get("sq", {}).get("SQ_INSTS_VALU", 0.0) * n
        active += e.get("sq", {}).get("SQ_ACTIVE_INST_VALU", 0.0) * n
    return us, traffic, insts, active


def frames_of(pmc):
    return max((e["launches"] for k, e in pmc.items() if k.startswith("me_fullpel_85pu_kernel")), default=0)


def stage_roofline(stage, ms, n_sb, pmc=None):
    """One stage of one frame: ms = its time (HIP events in bench.py, summed rocprofv3 durations in summarize_profiles.py)."""
    t = ms * 1e-3
    alg = ALG_BYTES_PER_SB[stage] * n_sb
    r = {"ms": ms, "algorithmic_bytes": alg, "algorithmic_GBps": alg / t / 1e9, "hbm_frac": alg / t / HBM_PEAK_BPS}
    if stage in USEFUL_LANE_OPS_PER_SB:
        useful = USEFUL_LANE_OPS_PER_SB[stage] * n_sb
        r.update({"useful_lane_ops": useful, "useful_frac": useful / t / VALU_PEAK_LANE_OPS, "useful_frac_of_measured_peak": useful / t / VALU_MEASURED_LANE_OPS})
    frames = frames_of(pmc) if pmc else 0
    if frames:
        us, traffic, insts, active = stage_counters(pmc, stage, frames)
        if traffic:
            r.update({"traffic_bytes": traffic, "traffic_over_algorithmic": traffic / alg})
        if insts:
            r.update({"issued_lane_ops": insts * 64.0, "issued_frac": insts * 64.0 / t / VALU_PEAK_LANE_OPS, "issued_frac_of_measured_peak": insts * 64.0 / t / VALU_MEASURED_LANE_OPS,
                      "profile_ms": us * 1e-3})
            if "useful_lane_ops" in r:
                r["issued_over_useful"] = insts * 64.0 / r["useful_lane_ops"]
        if active and us:
            r["valu_busy"] = active * 4.0 / (1024 * us * 1e-6 * 2.4e9)
        wait = sum(e.get("sq", {}).get("SQ_WAIT_ANY", 0.0) * e["launches"] for k, e in pmc.items() if any(k.startswith(p) for p in STAGE_KERNELS.get(stage, ())))
        cyc = sum(e.get("sq", {}).get("SQ_WAVE_CYCLES", 0.0) * e["launches"] for k, e in pmc.items() if any(k.startswith(p) for p in STAGE_KERNELS.get(stage, ())))
        if cyc:
            r["wait_any_over_wave_cycles"] = wait / cyc
    return r


def roofline(stage_ms, n_sb, pmc=None, pmc_source=None):
    """The bench line's `roofline` object: the dominant stage against what binds it + the same figures for every stage."""
    stages = {s: stage_roofline(s, ms, n_sb, pmc) for s, ms in stage_ms.items() if s in ALG_BYTES_PER_SB}
    dom = max(stages, key=lambda s: stages[s]["ms"])
    d = stages[dom]
    out = {"stage": dom, "kernel": "+".join(STAGE_KERNELS[dom]), "source": pmc_source}
    if "useful_frac" in d:        # a search: integer-VALU bound by construction.  frac = the USEFUL (algorithmic) share of the peak; what the ISA issued stands beside it
        out.update({"bound": "valu", "achieved": d["useful_lane_ops"] / (d["ms"] * 1e-3) / 1e12, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T lane-op/s", "frac": d["useful_frac"],
                    "useful_frac": d["useful_frac"], "issued_frac": d.get("issued_frac"), "issued_over_useful": d.get("issued_over_useful"), "valu_busy": d.get("valu_busy"),
                    "wait_any_over_wave_cycles": d.get("wait_any_over_wave_cycles"),
                    "measured_issue_peak": VALU_MEASURED_LANE_OPS / 1e12, "frac_of_measured_peak": d.get("useful_frac_of_measured_peak"),
                    "issued_frac_of_measured_peak": d.get("issued_frac_of_measured_peak")})
    else:
        out.update({"bound": "hbm", "achieved": d["algorithmic_GBps"], "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s", "frac": d["hbm_frac"]})
    out["traffic"] = d.get("traffic_bytes")
    out["traffic_over_algorithmic"] = d.get("traffic_over_algorithmic")
    out["stages"] = stages
    rf, wf, cal_src = counter_calibration()
    out["counter_calibration"] = {"fetch_size_factor": rf, "write_size_factor": wf, "source": cal_src}
    out["definitions"] = ("tools/roofline_defs.py: frac = useful_frac = the stage's algorithmic work count (USEFUL_LANE_OPS_PER_SB: one derivation, in code) / stage time / 78.6 T lane-op/s; "
                          "issued_frac = SQ_INSTS_VALU x 64 of the stage's kernels per frame (PMC pass of `source`) over the same time and peak; traffic = FETCH_SIZE x fetch_size_factor + "
                          "WRITE_SIZE x write_size_factor per frame (counter x 1024 x the factor measured by tools/calibrate_counters.sh); algorithmic bytes = SURVEY 8(d); "
                          "measured_issue_peak = what the chip issues of these integer / packed-16-bit instructions by wall clock (tools/ubench/valu_peak.hip, profiles/r06/valu_peak.txt): the *_of_measured_peak fields divide by it")
    return out
