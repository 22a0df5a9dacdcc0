// Run-time knobs of libc2w_hip.so -- ALL of them (DESIGN.md section 10 lists them with their purpose).  They override the dispatcher
// for tests and A/B measurements; none changes a result beyond the kernels' own rounding.  The environment is read ONCE, at the
// first launch; c2w_knobs_reload() (exported) re-reads it -- a test that flips a knob inside one process calls it afterwards.
#pragma once

struct C2wKnobs {
    bool force_gather;    // C2W_FORCE_GATHER=1   every conv / weight gradient on the general gather kernels (no halo-patch kernels)
    int conv_t3;          // C2W_CONV_T3          -1 (default): 16x16-tile conv kernel from conv_t3_min_wgs workgroups; 0: never; 16: wherever the image is tiled
    bool conv_pair;       // C2W_CONV_PAIR=0      8-pixel-wide images NOT paired on the halo-patch kernels (gather kernels instead)
    bool conv_ts2_patch;  // C2W_CONV_TS2_PATCH=0 stride-2 input gradient NOT on the parity-class halo-patch kernel
    int conv_s2_patch;    // C2W_CONV_S2_PATCH    1 (default): stride-2 FORWARD on the parity-plane halo-patch kernel where it pays (>= 4 K chunks or <= 2048 workgroups); 0: never (gather kernel, rounds 1-5); 2: wherever the geometry allows
    bool ts2_one_launch;  // C2W_TS2_FOUR_LAUNCHES=1 stride-2 input gradient as one launch per parity class instead of one launch for the four
    bool ts2_pairs;       // C2W_TS2_PAIRS=0      stride-2 input gradient with one class per workgroup (round 4) instead of two (round 6; 16-bit)
    bool up_patch;        // C2W_NO_UP_PATCH=1    up-convs NOT on the halo-patch kernels (upsampling folded into the gather kernel instead)
    bool pool2;           // C2W_NO_POOL2=1       c2w_conv_pool2_supported answers 0 (callers run conv + c2w_sumpool2)
    bool ln_fusion;       // C2W_NO_LN_FUSION=1   no LayerNorm forward / backward in conv epilogues (callers run the separate passes)
    bool lnf;             // C2W_NO_LNF=1         no LayerNorm FORWARD emission only
    bool wgrad_atomics;   // C2W_WGRAD_ATOMICS=1  split-K partial sums by fp32 atomics even when a workspace is handed over
    bool attn_valu;       // C2W_ATTN_VALU=1      attention on the fp32 VALU kernels instead of the matrix-core ones
    bool wgrad_narrow;    // C2W_NO_NARROW=1      edge convs (<= 80 output channels) NOT on the narrow forms of the halo-patch kernels
    bool wpacked;         // C2W_NO_WPACKED=1     c2w_conv_wpacked_supported answers 0 (callers hand over the plain [rows][9][Cin] weights)
    bool loss_fusion;     // C2W_NO_LOSS_FUSION=1 c2w_conv_loss_supported answers 0 (callers run the output conv and c2w_mse_loss_grad_noise)
    bool ln_chain;        // C2W_NO_LN_CHAIN=1    c2w_conv_lnfwd_chain_supported answers 0 (every residual block writes its output)
    int half8_max_wgs;    // C2W_HALF8_MAX_WGS=N  launches of up to N workgroups take the eight-wave 8x16-tile kernel (default 0 = automatic: every 16-bit launch, fp32 up to 256)
    bool half8;           // C2W_NO_HALF8=1       every 8x16-tile launch on the 4-wave kernel (rounds 1-5)
    bool half8_db;        // C2W_HALF8_DB=0       eight-wave launches of at most 256 workgroups with ONE patch buffer (an exposed patch load per K chunk)
    bool splitk;          // C2W_NO_SPLITK=1      c2w_conv_splitk_plan answers 1 (no convolution splits its K chunks over workgroups)
    int conv_t3_min_wgs;  // C2W_CONV_T3_MIN_WGS=N  workgroups from which the 16x16-tile conv kernel replaces the 8x16 one (default 512 = one round of two workgroups per CU; 1024 in rounds 1-5)
    int wgrad_wgs;        // C2W_WGRAD_WGS=N      workgroups a halo-patch weight-gradient launch splits its K range into (default 256: one per CU)
};

const C2wKnobs& c2w_knobs();
extern "C" void c2w_knobs_reload(void);
