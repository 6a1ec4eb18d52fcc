// smk_kernels.h -- launch parameter blocks + index helpers shared by host and device code.
// gfx950 only.  Activations are NHWC with the channel count padded to a multiple of 8;
// weights are packed [Npad][Kpad] (K-major), K ordered (kh, kw, cin_padded), BN folded.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SMK_HD __host__ __device__ inline
#else
#define SMK_HD inline
#endif

namespace smk {

// DT_F16X3 (round 6): the argmax-exact fp16 context.  Every value of the track path's trunk (stem .. cls / loc head.3) is a PAIR of fp16
// planes hi = fp16(v), lo = fp16(v - hi), and a product x * w is the three MFMA products x_hi w_hi + x_hi w_lo + x_lo w_hi in the fp32
// accumulator (x_lo w_lo, 2^-22 relative, is dropped).  Stored as THREE channel planes [hi | hi | lo] against weights packed [w_hi | w_lo |
// w_hi] per tap, a convolution is the SAME implicit GEMM with K tripled: the fp16 kernels run unchanged (the matrix pipe keeps fp16
// denormals, tools/measure/gpu_denorm_probe.py), only the epilogue splits.  Kernels see DT_F16; the engine's dtype says which packs exist.
enum { DT_F32 = 0, DT_F16 = 1, DT_F16X3 = 2 };
enum { RES_NONE = 0, RES_PRE_RELU = 1, RES_POST_RELU = 2 };
enum { OUT_NHWC = 0, OUT_NCHW_F32 = 1 };

// K tiles are 128 or 256 bytes of K per row (template parameter of the conv kernel)
constexpr int NPAD_ALIGN = 128;   // weight rows padded to the largest N tile
constexpr int KPAD_ALIGN = 128;   // elements; a multiple of every K-tile size (256 B of f16)

struct ConvParams {
    const void *in;        // NHWC storage tensor [B][Hs][Ws][Cs] (dtype)
    const void *wgt;       // [Npad][Kpad] (dtype)
    const void *wgt_frag;  // f16 only: the same matrix in MFMA-fragment order (conv_wreg_kernel), or nullptr:
                           //   [Npad/32][Kpad/16][64 lanes][8 halves], lane = (n % 32) + 32 * ((k % 16) / 8)
    const void *wgt_frag_halo;   // 3x3, f16: fragment order of the CHUNK-MAJOR matrix (K = (chunk of 64 channels, kh, kw, channel)), or nullptr
    const float *bias;     // [Npad] f32 (BN beta' or conv bias; zero padded)
    const void *res;       // optional residual, NHWC [B][Ho][Wo][res_Cs] (dtype)
    void *out;             // NHWC dtype  or  NCHW f32
    const int *pos;        // optional per-item (y,x) pairs selecting the window origin
    const void *zero;      // >= 8 KB of zeros in device memory (source of all zero padding)
    int B;
    int Hs, Ws, Cs;        // storage dims of the input, channel stride per pixel
    int cin_off;           // first channel of the slice that is read
    int Ci;                // logical input channels, multiple of 8; K = kh*kw*Ci
    int Hl, Wl;            // logical input image (window / upsampled view of the storage)
    int org_y, org_x;      // constant origin of the logical image inside the storage
    int pos_mul, pos_add;  // origin += pos*pos_mul + pos_add when pos != nullptr
    int ups;               // 1: logical image = nearest-upsampled storage (sy = ly*Hs/Hl)
    int Ho, Wo;
    int kh, kw, stride, pad, dil;   // stride = vertical stride; horizontal stride is stride_x
    int stride_x;
    int K, Kpad;
    int N;                 // real output channels
    int Nst;               // channels stored in NHWC mode (multiple of 4, >= N, zeros above N)
    int M;                 // B*Ho*Wo
    int Cos, cout_off;     // NHWC output channel stride / offset
    int res_Cs, res_coff;
    int relu, res_mode, out_mode;
    // grouped launch (blockIdx.z = group): per-group element offsets
    int groups;
    int g_cin_off;         // added to cin_off
    int g_wgt_off;         // rows of wgt/bias per group (multiple of NPAD_ALIGN)
    int g_cout_off;        // added to cout_off
    int xcd_mode;          // 0: tiles in launch order, 1: XCD-contiguous tm-major, 2: tn-major
    int ci_shift;          // log2(Ci) when Ci is a power of two, else -1 (division fallback)
    int kw_magic;          // tap / kw == (tap * kw_magic) >> 16  for tap < 4096
    int prio;              // s_setprio of the consumer waves (0..3); -1: producers at 1
    int wave_prio;         // s_setprio of EVERY wave of the launch (0..3; smk_tune main_prio): a pipelined step's main part above the previous frame's tail it shares SIMDs with
    int buf_lds;           // producers use buffer_load ... lds (SRD + 32-bit offsets, hardware zero fill)
    int a_stage;           // conv_wreg / conv_seq producers: 1 = activation rows global -> VGPR -> ds_write (A/B knob "a_stage")
    int res_nt;            // conv_wreg / conv_seq: residual rows fetched non-temporally (A/B knob "res_nt")
    unsigned in_bytes, w_bytes;   // extents of the input tensor / weight pack for the SRDs
    int nt_store;          // NCHW f32 epilogue: non-temporal stores (large tensors handed to the caller)
    // split-K across workgroups (NHWC epilogue): scratch for the f32 partial tiles and one arrival counter per
    // tile (zero between launches); ksplit is decided at launch (launch_conv_mfma_batch), 1 = off
    // DT_F16X3 contexts.  A split tensor is STORED as two channel planes [hi | lo] (hi = fp16(v), lo = fp16(v - hi)); the OPERAND a tripled-K
    // pack multiplies is [hi | hi | lo] (against [w_hi | w_lo | w_hi] per tap) -- the gather maps operand channel c of a tap to stored channel
    // c < x3_in ? c : c - x3_in (x3_in = the input's plane stride, 0 = not a split tensor; p.Ci stays the operand's 3 x3_in, p.Cs the stored 2 x3_in).
    // NHWC epilogues: x3_out > 0 = the output is a split tensor -- channel n goes to n (hi) and n + x3_out (lo = v - hi); x3_res > 0 = the residual
    // is one: value = res[n] + res[n + x3_res].  Plane strides in channels.
    int x3_out, x3_res, x3_in;
    // conv_wreg_kernel on a pack in FUSED order (wreg_tile.inc): x3_ct = C / 64 > 0 -- the fragment pack's weight tiles per tap are (w_hi_0, w_lo_0, w_hi_1, ..,
    // w_hi_0 .. for the lo plane), the activation gather is a plain one over the stored 2 C channels (the host hands the kernel Ci = 2 C, x3_in = 0: conv_wreg.hip
    // launch_wreg_t); x3_nreal = activation tiles the producers fetch (weight tiles that do not re-use their predecessor's)
    int x3_ct, x3_nreal;
    const float *oscale;   // DT_F16X3: [Npad] f32 or nullptr -- the accumulator of channel n is multiplied by oscale[n] in front of the bias.  The rows of a
                           // split pack are scaled by powers of two (max |w| of a row -> [2^13, 2^14)) so that w_lo stays a NORMAL fp16 number
                           // (unscaled, BN-folded weights of 1e-2 .. 1e-3 leave it subnormal: 3e-6 .. 3e-5 relative instead of 2^-22); oscale undoes it, exactly
    int ksplit;
    float *ks_part;
    unsigned *ks_cnt;
    size_t ks_part_cap;    // floats
    int ks_cnt_cap;
};

// several independent convolutions in one launch (same kernel instantiation for all of them)
constexpr int CONV_BATCH_MAX = 4;
struct ConvBatch {
    int n;                                // problems in this launch
    int start[CONV_BATCH_MAX + 1];        // first workgroup of problem i; start[n..] = grid size
    ConvParams p[CONV_BATCH_MAX];
};

// one layer of a persistent convolution sequence (conv_seq_kernel): the ConvParams fields conv_wreg's tile routine
// reads, under the same names, packed (104 bytes) so that layer2 + layer3 + adjust (33 convolutions) travel in ONE
// 4 KB kernel-argument segment
struct SeqLayer {
    const void *in;        // NHWC f16 [B][Hs][Ws][Cs]
    const void *wgt_frag;  // fragment-order weights
    const float *bias;
    const void *res;       // residual (NHWC f16) or nullptr
    void *out;             // NHWC f16
    unsigned in_bytes, w_bytes;
    int kw_magic;
    // 16-bit fields in 4-byte aligned PAIRS that are read together: the compiler fetches a pair with one scalar load (a pair that
    // straddles a dword boundary becomes a VECTOR load from the kernel-argument segment -- and a vmcnt wait in the producers)
    unsigned short Hs, Ws, Hl, Wl, Ho, Wo, Cs, cin_off, Ci, Kpad, Nst, Cos, res_Cs, res_coff, cout_off, bar_ord;
    // bar_ord: ordinal (1, 2, ...) of the team barrier BEHIND this layer, filled in by launch_conv_seq from `sync` (0 = none) --
    // the kernel does not carry a barrier count from layer to layer (it sat in a VGPR and was the one value it spilled)
    short org_y, org_x;
    signed char kh, kw, stride, stride_x, pad, dil, relu, res_mode, ci_shift;
    signed char cfg;       // workgroup tile: 0 = 64x256, 1 = 64x128, 2 = 64x64, 3 = 128x256, 4 = 128x128, 9 = 128x64;
                           //   measurement variants: 5 = 64x128 with a 5-deep ring and weights 4 K tiles ahead, 6..8 ablations
    signed char sync;      // 1: the next layer reads what this one (or an earlier one since the last barrier) wrote
    signed char a_stage;   // bit 0: see ConvParams::a_stage; bits 1-3 (engine.cpp seq_mark_resident, round 6): SEQ_YRES_IN / _NOSTORE / SEQ_LDS_HI
    signed char kstag;     // 1: every workgroup starts its K loop at another K tile (see wreg_tile kt0)
    signed char res_nt;    // 1: residual rows are fetched non-temporally (the tensor is dead after this layer)
    // features of ConvParams the sequences never use (compile-time constants for the shared tile routine)
    static constexpr const int *pos = nullptr;
    static constexpr int pos_mul = 0, pos_add = 0, ups = 0, g_cin_off = 0, g_wgt_off = 0, g_cout_off = 0;
    static constexpr int x3_out = 0, x3_res = 0, x3_in = 0, x3_ct = 0, x3_nreal = 0;   // (split-operand tensors never run inside a sequence)
    static constexpr const float *oscale = nullptr;
};
static_assert(sizeof(SeqLayer) == 104, "SeqLayer packing");
// pair codes (engine.cpp seq_fuse_pairs, c3c1_tile.inc): a Bottleneck's conv3 and the 1x1 convolution that reads its output run as
// ONE tile routine; the first record carries the shape code, the second one is consumed with it
constexpr int SEQ_CFG_C3C1_L3 = 20;    // K 256 -> N 1024 (+ residual, ReLU) -> N 256   (layer3 conv3 -> next conv1 / adjust)
constexpr int SEQ_CFG_C3C1_L2 = 21;    // K 128 -> N 512  (+ residual, ReLU) -> N 128   (layer2 conv3 -> next conv1)
constexpr int SEQ_CFG_C3C1_2ND = 22;   // the pair's second record
// the same pairs as a 2-D split over a PAIR of CUs (c3c1p_tile.inc): 64-row tiles, each CU half of conv3's channels and the
// matching K half of the second convolution, fp32 partial sums exchanged through SeqArgs::xch; second record = 22 as well
constexpr int SEQ_CFG_C3C1P_L3 = 26;
constexpr int SEQ_CFG_C3C1P_L2 = 27;
// triples (c3c1_tile.inc FRONT = 1; engine.cpp seq_fuse_triples): a Bottleneck's 3x3 convolution, its conv3 and the next 1x1 as ONE tile
// routine on image-row tiles; records: conv2 = 28 / 29, conv3 = 30, the 1x1 = 22
constexpr int SEQ_CFG_C2C3C1_L3 = 28;  // 3x3 256 -> 256, then the layer3 pair
constexpr int SEQ_CFG_C2C3C1_L2 = 29;  // 3x3 128 -> 128, then the layer2 pair
constexpr int SEQ_CFG_C2C3C1_MID = 30; // the triple's conv3 record (a_stage = 1: residual rows requested behind the team wait)
constexpr int SEQ_XCH_SLAB = 32768;    // bytes of one slab (c3c1p_tile.inc C3C1P_SLAB_MAX); a pair owns 2 sets x 2 destinations
constexpr int SEQ_XCH_PAIRS = 16;      // pairs per team the scratch is sized for (32 workgroups per XCD)
constexpr size_t SEQ_XCH_BYTES = (size_t)8 * SEQ_XCH_PAIRS * 4 * SEQ_XCH_SLAB;
// 3x3 stride-1 (dilated) convolutions on whole-row tiles with the activation patch shared by the nine taps (wreg_halo_tile.inc);
// the record's wgt_frag points at the chunk-major fragment pack (PackedConv::w_frag_halo)
constexpr int SEQ_CFG_HALO128 = 24;    // 128 pixels (whole output rows) x 64 channels
constexpr int SEQ_CFG_HALO64 = 25;     // 64 pixels x 64 channels
// Resident trunk (round 6): with one image per team a layer's fused pairs run on the SAME 32 rows in the SAME workgroup, Bottleneck after
// Bottleneck, so the pair's LDS image Y (relu(conv3 + residual), the next block's residual) stays where it is: the next pair does not
// fetch it back (SEQ_YRES_IN), this pair does not store it (SEQ_YRES_NOSTORE: nobody else reads the tensor), and the 3x3 convolution
// in between works in the LDS above it (SEQ_LDS_HI on its record; its 128-row epilogue then runs in two halves).
constexpr int SEQ_YRES_IN = 2, SEQ_YRES_NOSTORE = 4, SEQ_LDS_HI = 8;
constexpr int SEQ_YRES_BYTES = 65536;  // [32][1024] f16: the largest Y image (layer3); the routines between two pairs start here
constexpr int SEQ_MAX = 36;
struct SeqArgs {
    int n, B;
    int flags, pad_;       // flags bit 0: the team barrier polls through the scalar memory path (smk_tune "seq_spoll")
    unsigned *bar;         // [8 teams][32] u32, zero between launches: [0] barrier arrivals, [1] exits, [2] tickets, [8 + p] exchanges of pair p
    float *xch;            // SEQ_XCH_BYTES of scratch for the pair-split tiles' partial sums (nullptr: the list has none)
    int *err;              // device flag: 1 = an XCD received more workgroups than grid / 8, 2 = barrier timeout; a launch that
                           //   finds it set returns at once
    int *err_host;         // the same flag in host-mapped pinned memory (the engine checks it at every entry, no sync)
    unsigned long long *clk;   // optional [2 * SEQ_MAX + 1]: 100 MHz timestamps of (team 0, slot 0): start, then per layer
                               //   (tiles done, barrier passed) -- measurement aid (SMK_SEQ_CLK=1)
    unsigned long long *clk2;  // optional [8 * SEQ_MAX] (SMK_SEQ_CLK=2): per layer, the phases of (team 0, slot 0)'s first tile, see wreg_tile
    unsigned *exit_sem;        // optional (pipelined frame step, depth 2): the last team to leave adds one to exit_sem[0] (a semaphore the tail's second
                               //   part waits on); exit_sem[1] counts the teams that have left (zero between launches)
    SeqLayer L[SEQ_MAX];
};
static_assert(sizeof(SeqArgs) <= 4096, "the layer list travels in the kernel-argument segment");

// run-time tuning knobs (smk_tune): measured defaults, overridable for A/B runs
struct Tuning {
    int xcd_mode = 1;
    int force_tile = 0;        // 0 auto, 1 128x128, 2 128x64, 3 64x128, 4 64x64
    int min_blocks_x16 = 16;   // shrink tiles while grid < CUs * min_blocks_x16/16
    int stages = 0;            // LDS ring depth of the conv kernel: 0 auto, 2..4
    int kt = 0;                // K tile bytes: 0 auto, 128 or 256
    int prio = 0;              // consumer-wave priority (see ConvParams::prio)
    int nt_store = 1;          // non-temporal stores for NCHW outputs >= 4 MB (the 63x63 mask logits)
    int halo = 1;              // 3x3 stride-1 convolutions through conv3x3_halo_kernel (0 off, 1 per-shape choice,
                               // 128 / 64 force that workgroup height)
    int halo_db = 1;           // halo kernel: double-buffered patch when the launch has at most one workgroup per CU
    int chain = 1;             // fp16: Refine's sequential tail as one launch (refine_chain_kernel)
    int ksplit = 0;            // split-K across workgroups: 0 off (default: measured a net loss at B=8, +1 % at B=1,
                               // see pick_ksplit), 1 auto (long-K few-tile launches), 2 / 4 forced (tests)
    int xc_ch = 64;            // dw_xcorr, banded kernel: channels per workgroup (64 or 32)
    int l1_fused = 1;          // fp16: every layer1 Bottleneck as ONE launch with its weights in registers (l1_block_kernel); 0 = 3-4 launches
    int stem_fused = 1;        // fp16: cvt_in + stem + maxpool as ONE launch (stem_pool_kernel); 0 = the three launches of rounds 1-2
    int xc_full = 0;           // dw_xcorr: 0 = 5-row bands x 64 channels (480 small workgroups, input read 1.8x: 13.9 us at B = 8, the
                               // fastest -- default), 1 = 13-row bands (two per image, input read 1.14x, all loads of a thread in flight
                               // before the first LDS write: 20.5 us -- fewer, fatter workgroups lose the overlap), 2 = 5-row bands with
                               // the batched loads (21.5 us); rocprofv3, profiles/r03_dw_xcorr_variants.txt
    int buf_lds = 1;           // LDS-DMA through buffer resources instead of flat global addresses (measured
                               // faster: l3.0.ds 94 -> 76 us at B=8, profiles/r01_v5_ab_buf_lds.txt)
    int mask_overlap = 0;      // smk_step: mask head on a side stream beside decode + Refine (measured slower:
                               // a cross-stream graph edge makes hipGraphLaunch cost ~1 ms of host time)
    int chain_mask = 1;        // fused frame step, fp16: the mask head runs inside the Refine chain launch (chain_mask_kernel)
    int nchw_tn_major = 1;     // large NCHW f32 outputs (the 63x63 mask logits): tn-major tile order (see conv_params)
    int merge = 1;             // share one launch between independent convolutions (ds+c1, cls3+loc3, Refine windows): 0 never, 1 up to
                               // merge_max_batch streams (beyond it every member fills the chip by itself), 2 always
    int merge_max_batch = 32;
    int rf_wreg = 3;           // Refine's two merged front launches on the register-fed kernel's 64x64 tile: bit 0 the window convolutions + deconv, bit 1 the v*.2 launch
    int seq_fuse3 = 0;         // conv_seq_kernel: [conv2, conv3, next 1x1] of a Bottleneck as one tile routine on image-row tiles (0 off, 1 on, 2 layer3 only)
    int seq_spoll = 1;         // conv_seq_kernel's team barrier polls with s_load_dword glc (scalar path) instead of a vector sc1 load      // measured (profiles/r04k_merge_crossover_ab.txt): merging -5 % at B = 10, -1.5 % at B = 16, 0 at B = 24, +3.3 % at B = 32, +5.1 % at B = 64
    int wreg = 1;              // fp16 NHWC convolutions through conv_wreg_kernel (weights global -> VGPR, activations
                               // through LDS): 0 off, 1 per-shape choice (wreg_choice), 2..7 force tile code 1..6 where eligible
    int wreg_stages = 0;       // A-ring depth of conv_wreg_kernel: 0 auto (3), 3 or 4
    int seq = 1;               // fp16: ResNet layer2 .. adjust as ONE persistent per-XCD launch (conv_seq_kernel) for the batches below
    int seq_min_batch = 5, seq_max_batch = 8;      // batches that run layer2 .. adjust as the persistent sequence (engine.cpp seq_wanted)
    int seq_extra_batch = 12;                      // ... one more measured batch (four XCDs with two images, four with one: x1.029)
    int seq_mult_max = 24;                         // ... and the multiples of 8 up to this one
    int ablate = 0;            // MEASURE=1 builds only: conv_wreg_kernel with parts of the K loop removed (bits: conv_wreg.hip)
    int seq_tall = 2;          // sequences: 128-row tiles for layers that would otherwise need several 64-row rounds per image
                               // (1: short-K layers only -- the rule with two producer waves; 2: all, measured -1.7 % with four)
    int seq_kstag = 1;         // sequences: every workgroup of a team starts its K loop at another K tile (0 off, 1 layers whose
                               // weights fit the L2, 2 all)
    int res_nt = 1;            // conv_wreg / conv_seq: residual rows fetched non-temporally (the block input is dead after the add): -0.6 % B=8, -0.8 % B=64, bit-identical
    int seq_fuse = 1;          // sequences: a Bottleneck's conv3 + the next 1x1 convolution (next block's conv1 / adjust) as one tile routine
                               // on 32-row tiles (c3c1_tile.inc); 0 = two layers with a team barrier in between, 1 = every pair the routine has a
                               // shape for, 2 = layer3's pairs only (A/B knob), 3 = same as 1
    int rf_tile2 = 0;          // A/B knob: tile code (tile_from_code) of Refine's merged v*.2 launch, 0 = auto
    int pair_launch = 1;       // fp16, batches outside the persistent sequence: a Bottleneck's conv3 + the next 1x1 convolution as ONE launch
                               // (conv_pair_kernel = c3c1_tile per 32 rows of the flattened batch); 0 = two launches, 1 = the measured rule (off for B <= 2, 32-row tiles to B = 8, 64-row tiles -- c3c1s_tile -- from B = 9), 2 / 3 = always 32 / 64 rows
    int corr_head = 1;         // fp16: dw_xcorr + head.0 + cls / loc head.3 as ONE launch (corr_head.hip); 0 = the three launches of rounds 1-3
    int seq_yres = 1;          // sequences, batches of at most 8 (one image per team): the fused pairs' trunk image Y stays in LDS from Bottleneck to
                               // Bottleneck (no 64 KB re-fetch per CU and pair, no store of a tensor only the same workgroup reads); bit-identical
    int seq_search = 0;        // sequences (search branch): conv_search (N-fused, 128 x 256 tiles) as the persistent launch's LAST record instead of a launch
                               // of its own behind it.  Measured a wash (0.5499-0.5521 against 0.5501-0.5505 ms per B = 8 step: 34 us inside the sequence
                               // for the 36 us launch, profiles/r06i_seq_search_ab.txt): off
    int seq_pair2d = 0;        // sequences: the fused (conv3, next 1x1) pairs as a 2-D split over a PAIR of CUs (c3c1p_tile.inc: 64-row tiles,
                               // each CU half of conv3's channels + the matching K half of the second convolution, fp32 partial sums
                               // exchanged): 0 = c3c1_tile (one CU, 32 rows, all channels), 1 = every pair, 2 = layer3's pairs only
    int seq_halo = 1;          // sequences: 3x3 stride-1 layers with N <= 256 (the Bottlenecks' conv2) on whole-row tiles with the activation patch
                               // shared by the nine taps (wreg_halo_tile.inc); 0 = the im2col tiles of wreg_tile
    int seq_kstag_mask = 7;    // which tile routines of the sequences stagger their K loops: 1 = fused pairs, 2 = patch-sharing tiles, 4 = im2col tiles
    int seq_ds128 = 0;         // sequences: N = 512 long-K layers whose 64x256 tiling gives exactly one round (layer2.0's shortcut) on 128x128 tiles
    int seq_deep = 0;          // measurement: 64x128 sequence tiles with a 5-deep activation ring, weights four K tiles ahead
    int seq_first_stage = 1;   // first ResNet stage (0..2) inside the sequences; 3 = adjust only
    int wreg_policy = 1;       // which layers conv_wreg_kernel takes under wreg = 1: 0 = the round-2 table (fitted with two producer
                               // waves), 1 = the rule fitted with four (wreg_choice)
    int npw = 4;               // conv_wreg / conv_seq: producer waves per workgroup (2 or 4; measured: profiles/r02_producer_waves_2_vs_4.txt)
    int a_stage = 0;           // conv_wreg / conv_seq: activation rows through registers instead of LDS-DMA (see ConvParams::a_stage)
    int pp = 1;                // fp16 NHWC convolutions with M >= 32768 rows, K >= 2304 and >= 200 tiles of 256 x 256 through conv_pp_kernel (0 off, 1 the
                               // rule in engine.cpp pp_choice, 2 wherever eligible)
    int main_prio = 3;         // pipelined steps: wave priority (s_setprio 0..3) of the MAIN part's launches (stem_pool, l1_block, the per-layer convolutions) -- they share
                               // SIMDs with the previous frame's tail, whose launches stay at 0: B = 1 +1.2 %, f16x3 +1.4 .. 2.4 %, B = 8 / 64 unchanged (profiles/r06bb_main_part_wave_priority.txt)
    int front_occ1 = 0;        // MEASURE builds: bit 0 l1_block_kernel, bit 1 stem_pool_kernel limited to ONE workgroup per CU (padding LDS): does the pipelined
                               // step's tail run BESIDE the next frame's front end then?  No: +3.5 % per step (profiles/r06g_front_occupancy_ab.txt)
    int wreg96 = 1;            // conv_wreg tile choice: 96 x 256 tiles where 128 x 256 would leave a partial round (see wreg_choice)
    int x3_fused = 1;          // split-operand contexts: conv_wreg_kernel's fragment packs in fused order (a hi activation tile staged once for its two products)
    int wreg32 = 140;          // conv_wreg tile choice: 32 x 64 tiles where fewer than this many 64 x 64 tiles exist (0 = never): -8..15 % per under-filled
                               // launch, -2 % on the B = 1 step (profiles/r06w_wreg_32_row_tiles.txt)
    int pipe_join = 1;         // pipelined frame step: 1 = the join with the previous frame's tail is an in-stream gate kernel (two graphs per
                               // frame), 0 = a cross-queue event wait (three graphs; measured 15-22 us of latency on the critical path)
    int pipe_two_form = 1;     // depth-2 pipelining: 1 = the mask head as its own launch at the head of the tail's first part, the bare Refine chain beside
                               // the next frame's heads; 0 = chain + mask head as one launch beside the heads (measured: it starves conv_search);
                               // 2 = chain + mask head as one launch behind conv_search (beside corr_head + decode: 136-216 idle CUs)
    int pipe_prio = 0;         // (MEASURE builds) queue priority of the pipelined step's side stream (0 default, 1 lowest, 2 highest): read when the stream is created
    int pipe_late = 1;         // pipelined frame step outside the persistent sequence's batches: the main gate in front of the heads instead of in front of
                               // layer2 (the tail overlaps the whole backbone of the next frame; p2 exists twice as well)
    int pipe_sig = 2;          // pipelined frame step, how the side stream learns that decode(f) is done: 2 (default) = a one-wave gate kernel at the head of
                               // the tail polls a semaphore the decode launch's last writer raises (it polls through the next frame's persistent launch,
                               // beside it: 0.531-0.541 ms per step against 0.556 for 0 and 0.563-0.565 serial, profiles/r05f_pipe_sig_ab.txt);
                               // 0 = hipEventRecord in the step's stream + hipStreamWaitEvent; 1 = hipStreamWaitValue32 on signal memory (3-4 us in the
                               // two-kernel probe, 0.85 ms per step in the real loop, profiles/r05e_pipe_sig_ab.txt)
    int pipe_eager = 0;        // pipelined frame step, A/B knob: bit 0 = the front end (stem + layer1) as eager launches instead of a graph,
                               // bit 1 = the Refine / mask tail as eager launches
};
extern Tuning g_tune;

// decode a K index (start of a 16-byte vector) into tap + channel
struct KDecode { int kh_i, kw_i, c; };

SMK_HD KDecode decode_k(int kvec, int Ci, int kw) {
    KDecode d;
    int tap = kvec / Ci;
    d.c = kvec - tap * Ci;
    d.kh_i = tap / kw;
    d.kw_i = tap - d.kh_i * kw;
    return d;
}

// per output row (b, oy, ox): logical top-left input coordinate and window origin
struct RowInfo { int b, ly0, lx0, oy_org, ox_org; };

SMK_HD RowInfo row_info(const ConvParams &p, int m, const int *pos) {
    RowInfo r;
    int hw = p.Ho * p.Wo;
    r.b = m / hw;
    int rem = m - r.b * hw;
    int oy = rem / p.Wo;
    int ox = rem - oy * p.Wo;
    r.ly0 = oy * p.stride - p.pad;
    r.lx0 = ox * p.stride_x - p.pad;
    r.oy_org = p.org_y;
    r.ox_org = p.org_x;
    if (pos) {
        r.oy_org += pos[2 * r.b + 0] * p.pos_mul + p.pos_add;
        r.ox_org += pos[2 * r.b + 1] * p.pos_mul + p.pos_add;
    }
    return r;
}

// element offset of input vector (row r, tap d) inside p.in, or -1 when it is zero padding
SMK_HD long gather_offset(const ConvParams &p, const RowInfo &r, const KDecode &d, int cin_off) {
    int ly = r.ly0 + d.kh_i * p.dil;
    int lx = r.lx0 + d.kw_i * p.dil;
    if ((unsigned)ly >= (unsigned)p.Hl || (unsigned)lx >= (unsigned)p.Wl) return -1;
    int sy, sx;
    if (p.ups) {
        sy = (ly * p.Hs) / p.Hl;
        sx = (lx * p.Ws) / p.Wl;
    } else {
        sy = ly + r.oy_org;
        sx = lx + r.ox_org;
    }
    if ((unsigned)sy >= (unsigned)p.Hs || (unsigned)sx >= (unsigned)p.Ws) return -1;
    return ((long)(r.b * p.Hs + sy) * p.Ws + sx) * p.Cs + cin_off + d.c;
}

struct XcorrParams {
    const void *x;    // [B][H][W][Cs]   (conv_search output, all branches side by side)
    const void *k;    // [B][kh][kw][Cs] (cached conv_kernel(zf))
    void *out;        // [B][Ho][Wo][Cs]
    int B, H, W, kh, kw, Ho, Wo, C, Cs;
};

// corr_head.hip: dw-xcorr + head.0 + (cls / loc) head.3 as one launch, f16; tensors NHWC with channel stride Cs (branches side by side)
struct CorrHeadParams {
    const _Float16 *xs;        // [B][29][29][Cs]  conv_search output
    const _Float16 *zk;        // [B][5][5][Cs]    cached conv_kernel(zf)
    _Float16 *corr;            // [B][25][25][Cs]  out: correlation (corr_feature = its mask third)
    _Float16 *h0;              // [B][25][25][Cs]  out: head.0 output
    const void *w0_frag;       // head.0 pack in MFMA-fragment order, groups (branches) side by side: [nb * 8 blocks][16 k-steps][64][8]
    const float *b0;           // [nb * 256]
    const void *w3_frag[2];    // cls / loc head.3 packs in fragment order (block 0 holds the real rows), or nullptr
    const float *b3[2];
    float *out3[2];            // cls [B][10][25][25], loc [B][20][25][25] f32 NCHW
    int n3[2];
    unsigned w0_bytes, w3_bytes[2];
    int B, nb, Cs;
    unsigned *start_sem;       // optional: a semaphore raised once when the launch starts (pipelined frame step, depth 2)
};
int launch_corr_head(const CorrHeadParams &p, void *stream);

#ifdef __HIPCC__
__device__ __forceinline__ void set_wave_prio(int prio) {        // (s_setprio takes an immediate)
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
}
#endif
// DT_F16X3: a split tensor is stored as X3_PLANES channel planes [hi | lo] (the operand of a tripled-K pack is [hi | hi | lo]: ConvParams::x3_in)
constexpr int X3_PLANES = 2;
struct PoolParams { const void *in; void *out; int B, H, W, C, Ho, Wo; int x3; };     // x3: C channels stored as [hi | lo] planes (2 C per pixel)
// the fused stem (stem_pool.hip): NCHW f32 frame -> conv1 7x7/2 + BN + ReLU -> p0 [B][s0][s0][64] -> maxpool 3x3/2 p1 -> x1 [B][s1][s1][64], f16
struct StemPoolParams { const float *in; const void *wgt_frag; const float *bias; void *p0; void *x1; int B, S, s0, s1, Kpad; int prio; };   // prio: wave priority (s_setprio)

struct CvtInParams { const float *in; void *out; int B, C, H, W, Cpad; int pairs; int x3; };  // NCHW f32 -> NHWC dtype (x3: [hi | lo] planes of Cpad channels)
// pairs = 1 (stem input, C <= 4): [B][H][ceil(W/2)][2 pixels x 4 channels], missing pixel / channel = 0
struct CvtOutParams { const void *in; float *out; int B, C, H, W, Cs, coff; int plane; };  // NHWC dtype -> NCHW f32 (plane > 0: split tensor, value = hi + lo at coff + plane)

// on-device restatement of the host decode of tools/test.py:205-254 (one workgroup per stream)
struct DecodeParams {
    const float *cls;        // [B][2*A][S][S] f32 NCHW
    const float *loc;        // [B][4*A][S][S]
    const double *target_wh; // [B][2] f64 target size in crop pixels (w, h) = target_sz * scale_x (tools/test.py:230)
    const double *window;    // [S*S] cosine window (outer(hanning, hanning))
    int *pos_out;            // [B][2] (y, x) of the best anchor position
    double *box_out;         // [B][8] f64: cx, cy, w, h (crop pixels; float32 values), score (float32 value), penalty, pscore, best_id
    // scratch for the cross-workgroup argmax: winners per (stream, anchor shape); `arrived` starts at 0
    double *part_val;        // [B][8]
    int *part_idx;           // [B][8]
    double *part_box;        // [B][8][8]
    unsigned *arrived;       // [B]
    int B, A, S, stride;
    float anchor_w[8], anchor_h[8];
    double penalty_k, window_influence;
    // result ring (smk_set_result_ring), folded into this launch: the stream's final writer also stores the box into row
    // (*ring_cursor % ring_rows); ring_advance: no Refine launch follows, the last stream's writer advances the cursor
    double *ring_box;        // [rows][B][8] or nullptr
    int *ring_cursor;
    unsigned *ring_done;
    int ring_rows, ring_advance;
    int *ring_also;          // a second cursor advanced together with ring_cursor (the engine keeps a box-row cursor and a frames-committed cursor), or nullptr
    // pipelined frame step: the last stream's writer adds one to *mark (the semaphore the tail's gate waits on); mark_arrived counts streams
    unsigned *mark, *mark_arrived;
};

// result ring (misc_kernels.hip ring_commit_kernel): row (cursor % rows) <- this frame's box + fp16 Refine logits; cursor advances
struct RingParams {
    const double *box;         // [B][8] f64 (smk_step's box_out)
    const float *ref;          // [B][n] f32 (smk_step's refine_out) or nullptr
    double *box_ring;          // [rows][B][8]
    _Float16 *ref_ring;        // [rows][B][n] or nullptr
    int *cursor;               // device: frames committed so far
    unsigned *done;            // device: arrival counter of the launch's workgroups (zero between launches)
    int rows, B, n;
};

// image ops either side of the network (image_kernels.hip); per-stream scalars travel in the kernarg
constexpr int CROP_MAX_B = 32;
struct CropParams {
    const unsigned char *frames;   // [B or 1][H][W][3] uint8 (BGR as cv2.imread gives it)
    long frame_stride;             // bytes between per-stream frames (0: all streams share one frame)
    float *out;                    // [B][3][model_sz][model_sz] f32
    int H, W, model_sz;
    int box[CROP_MAX_B][3];        // xmin, ymin, sz  (un-padded frame coordinates)
    unsigned char avg[CROP_MAX_B][4];
};
struct PasteParams {
    const float *logits;           // [B][ms*ms] refine mask logits
    unsigned char *mask_out;       // [B][H][W] (prob > seg_thr) or nullptr
    float *prob_out;               // [B][H][W] warped probability or nullptr
    int ms, W, H;
    float seg_thr, border;
    double inv_map[CROP_MAX_B][6]; // inverse affine map (dst -> src), row major 2x3
};

// ---- launchers (defined in the .hip files) ---------------------------------------------
struct TileChoice { int bm, bn, kt, stages; };
TileChoice choose_tile(const ConvParams &p, int dtype);
int launch_conv_mfma(const ConvParams &p, int dtype, TileChoice t, void *stream);
int launch_conv_mfma_batch(ConvBatch &cb, int dtype, TileChoice t, void *stream);
int launch_conv_naive(const ConvParams &p, int dtype, void *stream);
// the split-K factor launch_conv_mfma_batch would use for this problem with this tile (1 = none)
int conv_ksplit(const ConvParams &p, int dtype, const TileChoice &t);
// 3x3 stride-1 convolution with the activation patch shared by the nine taps (chunk-major weight pack);
// returns 1 when the geometry is not eligible
int launch_conv_halo(const ConvParams &p, int dtype, int bm, void *stream);
// weights straight into registers (conv_wreg.hip): f16, NHWC epilogue, p.wgt_frag set; tile bm in {64,128} x bn in
// {64,128,256}; returns 1 when a problem of the batch is not eligible
bool conv_wreg_eligible(const ConvParams &p, int dtype);
int launch_conv_wreg_batch(ConvBatch &cb, int bm, int bn, int stages, void *stream);
// 256 x 256 tiles, eight waves in two alternating groups, both operands through LDS (conv_pp.hip): the long-K convolutions of
// the large-batch regime; returns 1 when the problem is not eligible
bool conv_pp_eligible(const ConvParams &p, int dtype);
int launch_conv_pp(const ConvParams &p, void *stream);
// a sequence of convolutions as one persistent launch of `grid` workgroups (one per CU, a multiple of 8)
int launch_conv_seq(const SeqArgs &a, int grid, void *stream);
// ONE fused (conv3 + residual + ReLU, next 1x1) pair as its own launch over M = B * H * W rows (code: SEQ_CFG_C3C1_L3 / _L2)
// rows: 32 (c3c1_tile) or 64 (c3c1s_tile: half the weight bytes per row, for large batches)
int launch_conv_pair(const SeqLayer &L3, const SeqLayer &L1, int code, int M, void *stream, int rows);
// resident workgroups per CU the runtime promises for conv_seq_kernel (0: it cannot run; smk_create's gate)
int conv_seq_occupancy();
// XCD id of every block of a `grid`-block launch -> host array (synchronous; smk_create's placement check)
int xcc_census(int grid, int *out_host);
// the sequential tail of Refine (h2, post0, h1, post1, h0, post2) as one launch, fp16 only (refine_chain.hip)
struct RefineChainLayer {
    const void *w;         // packed weights [rows][Kpad] fp16, K = (tap, channel of a Ci-channel image)
    const float *bias;
    int Kpad, Ci;
};
struct RefineChainParams {
    const void *d;                        // deconv output [B][15*15][32]
    const void *v2, *v1, *v0;             // ReLU(v*.2(...)) NHWC: [B][15*15][v2_cs], [B][31*31][v1_cs], [B][61*61][v0_cs]
    int v2_cs, v1_cs, v0_cs;
    RefineChainLayer L[9];                // h2.0 h2.2 post0 h1.0 h1.2 post1 h0.0 h0.2 post2
    float *out;                           // [B][127*127] f32
    int B;
    unsigned long long *clk;              // optional [11]: 100 MHz timestamps of workgroup 0 at the layer boundaries
    // result ring folded into the chain: post2 also stores fp16 logits into row (*ring_cursor % ring_rows); the last workgroup
    // to finish advances the cursor (every workgroup reads it at its start, the advance needs every workgroup's arrival)
    _Float16 *ring;                       // [rows][B][127*127] or nullptr
    int *ring_cursor;
    unsigned *ring_done;
    int ring_rows;
    // pipelined frame step, chain_mask_kernel only: the context's counter block (engine pipe_cnt); the launch's last workgroup adds one
    // to [0] (the tail semaphore), [5] counts the workgroups' arrivals; nullptr = not the end of a pipelined tail
    unsigned *tail_sem;
};
int launch_refine_chain(const RefineChainParams &p, void *stream);
// the chain and ONE NCHW f32 convolution (the mask head, 128x128 tiles) as one horizontally fused launch
int launch_chain_mask(const RefineChainParams &rp, ConvBatch &cb, void *stream);
int launch_xcorr(const XcorrParams &p, int dtype, void *stream);
// DT_F16X3 (x3_kernels.hip): x [B][H][W][2 Cx], k [B][kh][kw][2 Cx] in whole-tensor planes [hi | lo] (stride Cx = p.Cs), C = channels computed; out
// [B][Ho][Wo][2 Cx] PER-BRANCH planes: channel c = 256 g + cc lives at 512 g + 256 plane + cc (head.0 is a grouped convolution)
int launch_xcorr_x3(const XcorrParams &p, void *stream);
void xcorr_prepare();      // one-time kernel attribute set-up (large dynamic LDS); call outside stream capture
int launch_maxpool(const PoolParams &p, int dtype, void *stream);
int launch_stem_pool(const StemPoolParams &p, void *stream);
// one layer1 Bottleneck as one launch (l1_block.hip); w* = f16 packs in 16x16x32-fragment order (PackedConv::w_frag16),
// wd / bd = the 1x1 projection shortcut of block 0
struct L1BlockParams {
    const void *x; void *y;
    const void *w1, *w2, *w3, *wd;
    const float *b1, *b2, *b3, *bd;
    int B, S, Cin, K1pad, K2pad, K3pad, Kdpad;
    unsigned long long *clk;           // optional [6]: phase stamps of workgroup 0 (SMK_L1_CLK=1)
    int prio;                          // wave priority (s_setprio)
};
int launch_l1_block(const L1BlockParams &p, void *stream);
int launch_cvt_in(const CvtInParams &p, int dtype, void *stream);
int launch_cvt_out(const CvtOutParams &p, int dtype, void *stream);
int launch_cvt_in_x3(const CvtInParams &p, void *stream);       // x3_kernels.hip
int launch_cvt_out_x3(const CvtOutParams &p, void *stream);
int launch_maxpool_x3(const PoolParams &p, void *stream);
int launch_decode(const DecodeParams &p, void *stream);
int launch_ring_commit(const RingParams &p, void *stream);
// pipelined frame steps: in-stream gate (waits for the previous frame's tail) / the tail's completion mark; cnt = device [2] u32
// P: poll until *sem > 0, take one; gives up after 0.2 s (long_wait: 5 s, counted from the moment *started_wait is non-zero); then *started_wait = 0, *started_set = 1
int launch_pipe_gate(unsigned *sem, int *err, int *err_host, void *stream, int long_wait = 0, unsigned *started_wait = nullptr, unsigned *started_set = nullptr);
int launch_pipe_done(unsigned *sem, void *stream);                             // V
int launch_pipe_mark(unsigned *sig, void *stream);          // sig: signal memory (hipMallocSignalMemory), waited for with hipStreamWaitValue32
int launch_crop_resize(const CropParams &p, int B, void *stream);
int launch_paste_mask(const PasteParams &p, int B, void *stream);
int launch_paste_labels(const PasteParams &p, int n_obj, void *stream);
const void *zero_page();   // device-resident 8 KB of zeros (allocated on first use, per device)

}  // namespace smk
