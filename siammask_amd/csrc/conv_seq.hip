// conv_seq.hip -- conv_seq_kernel: a sequence of convolutions (ResNet stages: Bottleneck after Bottleneck,
// experiments/siammask_sharp/resnet.py:64-103,159-165) as ONE persistent launch.  Tile routine: wreg_tile.inc.
//
// Why: at B = 8 the step is ~50 dependent launches of 7-20 us.  Every launch boundary costs 1.5-2 us plus the
// write-back of what the predecessor left dirty (B / 6 TB/s), the grid fill / drain and each workgroup's cold start,
// and the next layer re-reads its input from the fabric because it was produced under other XCDs' L2s.  MI355X is
// eight XCDs with a private 4 MB L2 each -- and the workload is B independent images.  So: image b belongs to XCD
// b % 8 for the WHOLE sequence.  The 32 workgroups of an XCD (one per CU; a workgroup reads its XCD from
// HW_REG_XCC_ID and draws a ticket inside the team) share the tiles of their images layer by layer; between dependent
// layers they meet at a TEAM-LOCAL barrier: plain stores (they stay in the XCD's L2) -> s_waitcnt vmcnt(0) -> one
// L2-executed atomic per workgroup (ARRIVE) ... sc1 polls (WAIT).  No agent-scope release / acquire, no L2 write-back, no
// L1 invalidate: the consumers read the handed-over activations with sc1 loads (L2-served), and a layer's 2 MB of
// activations are L2 hits for the next one.  Weights are read-only (plain loads).  The kernel boundary at the end
// publishes the results to everybody else.
//
// Round 3: the barrier is SPLIT.  A workgroup arrives as soon as its stores have drained, then runs the part of the next
// layer that does not depend on the previous one -- layer decode, the consumers' first weight fragments (two K tiles),
// the producers' row / tap decode -- and only then waits (wreg_tile's hoist point), so the weight first touch and the
// address arithmetic overlap the barrier instead of following it.  The whole layer list (<= 36 layers: layer2 + layer3 +
// adjust = 33) travels in one kernel-argument segment: one launch per step instead of two.
//
// Failure is loud: a barrier that does not complete within 0.2 s (co-residency broken by a neighbour that holds CUs
// forever, or a second persistent kernel) and a team that received more workgroups than grid / 8 set a flag in device
// AND host-mapped memory, every workgroup abandons the remaining layers, a later launch that finds the flag set returns
// at once, and the engine turns the flag into an error at the next entry point (engine.cpp seq_health).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "smk_kernels.h"

namespace smk {

#include "wreg_tile.inc"
#include "c3c1_tile.inc"
#ifdef SMK_MEASURE
#include "c3c1p_tile.inc"      // (the pair split over two CUs: built, parity-green, a wash -- measurement builds only, see launch_conv_seq)
#endif
#include "c3c1s_tile.inc"
#include "wreg_halo_tile.inc"

// weight ring depth (k-steps in flight per consumer wave) of the two patch-sharing tiles, and one (SB = 1) or two sets of activation
// fragments (measured: tools/measure/gpu_halo_probe.sh builds the other combinations with -DSMK_HALO_D128=.. -DSMK_HALO_SB128=..)
#ifndef SMK_HALO_D128
#define SMK_HALO_D128 3      // (3 = 6 in time, profiles/r03h_halo_first_contact.txt: the loop is not latency-bound; 6 costs 24 registers and spills)
#endif
#ifndef SMK_HALO_SB128
#define SMK_HALO_SB128 0
#endif
constexpr int HALO_D128 = SMK_HALO_D128, HALO_SB128 = SMK_HALO_SB128, HALO_D64 = 6;
constexpr int SEQ_POLL_TID = 256;              // lane 0 of the first producer wave: it has no loads in flight at the hoist point
constexpr int SEQ_CLK2_STRIDE = 12;            // u64 per layer of the SMK_SEQ_CLK=2 stamps

__device__ __forceinline__ void seq_raise(const SeqArgs &a, int code) {
    __hip_atomic_store(a.err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.err_host) __hip_atomic_store(a.err_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ARRIVE: every wave's stores have reached the L2, then one atomic per workgroup
__device__ __forceinline__ void team_arrive(unsigned *cnt) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (threadIdx.x == SEQ_POLL_TID)
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // executes in the XCD's L2
}

// One read of a team counter.  spoll: through the SCALAR path (s_load_dword ... glc: misses the scalar cache, served by the XCD's L2,
// where the arrivals' atomics execute) -- the poll then does not travel the CU's vector memory path, where the workgroup's own requests
// issued in front of the wait are queued (HISTORY.md 3.1n); otherwise a vector sc1 load.  SeqArgs::flags bit 0 (smk_tune "seq_spoll").
__device__ __forceinline__ unsigned team_poll(const unsigned *cnt, bool spoll) {
    if (spoll) {
        unsigned v;
        asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(cnt) : "memory");
        return v;
    }
    return __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1 load: L2-served
}

// WAIT, at wreg_tile's hoist point.  The consumer waves carry weight loads in flight there, so the workgroup meets at an
// LDS-only barrier (__syncthreads() would drain vmcnt).
struct TeamWait {
    const SeqArgs *a;
    unsigned *cnt;
    unsigned target;            // 0: nothing to wait for
    int *abort_sh;              // LDS flag: the poller gave up
    __device__ __forceinline__ bool operator()() const {
        if (target == 0) return true;
        if (threadIdx.x == SEQ_POLL_TID) {
            const unsigned long long t0 = wall_clock64();
            while (team_poll(cnt, (a->flags & 1) != 0) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 20000000ull) {               // 0.2 s at 100 MHz: never hang the GPU
                    seq_raise(*a, 2);
                    *abort_sh = 1;
                    break;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // (an LDS read: through the generic pointer this was a FLAT load, whose s_waitcnt vmcnt(0) drained the weight fragments
        //  every consumer wave has in flight here)
        return *(__attribute__((address_space(3))) volatile int *)abort_sh == 0;
    }
};

// T3 = 1: the instantiation that also carries the triple routine (c3c1_tile FRONT = 1), launched only for lists that hold a triple --
// carrying it costs every other routine of the kernel (196 against 100 spilled SGPRs: +7 us on the B = 8 sequence, profiles/r04x_*)
template <int NPW, int CLK = 0, int T3 = 0>
__global__ __launch_bounds__((4 + NPW) * 64, 1) void conv_seq_kernel(const SeqArgs a) {
    // (resident trunk: the patch-sharing 3x3 tile between two pairs works behind the pair's Y image -- two patch buffers, which also
    //  hold its 64-row accumulator hand-over)
    constexpr int SEQ_LDS0 = (!T3 || WregLds<4, 3>::v > C3C1Lds<256, 1024, 256>::vx) ? WregLds<4, 3>::v : C3C1Lds<256, 1024, 256>::vx;
    constexpr int SEQ_LDS = SEQ_LDS0 > SEQ_YRES_BYTES + 2 * HALO_PB ? SEQ_LDS0 : SEQ_YRES_BYTES + 2 * HALO_PB;
    static_assert(4 * 64 * 68 * 4 <= 2 * HALO_PB && C3C1Lds<256, 1024, 256>::A_OFF == SEQ_YRES_BYTES && C3C1Lds<128, 512, 128>::A_OFF <= SEQ_YRES_BYTES, "resident Y");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SEQ_LDS];
    static_assert(C3C1Lds<256, 1024, 256>::v <= WregLds<4, 3>::v && C3C1Lds<128, 512, 128>::vx <= SEQ_LDS && SEQ_LDS + 64 <= 160 * 1024, "LDS of the fused tiles");
    static_assert(NPW == 4, "c3c1_tile computes on all eight waves");
    __shared__ int ctl[4];                               // [0] slot, [1] error flag found at entry, [2] abort
    // team = the XCD this workgroup really runs on (HW_REG_XCC_ID; the dispatcher deals consecutive blocks round-robin
    // over the XCDs, starting wherever the previous launch stopped, so blockIdx says nothing); slot = arrival ticket
    // inside the team.  A one-block-per-CU launch puts gridDim/8 workgroups on every XCD (checked at smk_create).
    const int nslots = gridDim.x >> 3;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int team = (int)(xcc & 7);
    unsigned *cnt = a.bar + team * 32;                   // one 128-byte line per team: [0] barrier, [1] exits, [2] tickets
    if (threadIdx.x == 0) {                              // two independent round trips, issued together
        const unsigned t = __hip_atomic_fetch_add(cnt + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int e = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ctl[0] = (int)t;
        ctl[1] = e;
        ctl[2] = 0;
    }
    __syncthreads();
    const int slot = ctl[0];
    if (ctl[1]) return;                                  // an earlier launch failed: do nothing until the host has dealt with it
    if (slot >= nslots) {                                // more workgroups on this XCD than the census promised
        if (threadIdx.x == 0) seq_raise(a, 1);
        return;
    }
    unsigned pending = 0;                                // arrival count of the barrier arrived at and not yet waited for
    const bool clk = CLK != 0 && a.clk && team == 0 && slot == 0 && threadIdx.x == 0;     // (the per-layer stamps live in the CLK build as well: SMK_SEQ_CLK=1 / 2 both launch it)
    if (clk) a.clk[0] = wall_clock64();
    bool alive = true;
    for (int li = 0; li < a.n && alive; ++li) {
        const SeqLayer &L = a.L[li];
        const int cfg = L.cfg;
        // cfg 20 / 21: this layer (a Bottleneck's conv3) and the NEXT record (the 1x1 convolution that reads it: cfg 22) run as
        // ONE tile routine on 32-row tiles, no barrier in between (c3c1_tile.inc; the engine's seq_fuse_pairs marks the pairs)
        const bool fusedp = cfg == SEQ_CFG_C3C1P_L3 || cfg == SEQ_CFG_C3C1P_L2;      // ... the same pair split over two CUs (c3c1p_tile.inc)
        const bool fused = cfg == SEQ_CFG_C3C1_L3 || cfg == SEQ_CFG_C3C1_L2;
        // cfg 28 / 29: this layer (a Bottleneck's 3x3 convolution), the NEXT record (its conv3: cfg 30) and the one behind it (the 1x1
        // that reads conv3's output: cfg 22) run as ONE tile routine on image-row tiles (c3c1_tile.inc, FRONT = 1; seq_fuse_triples)
        const bool fused3 = cfg == SEQ_CFG_C2C3C1_L3 || cfg == SEQ_CFG_C2C3C1_L2;
        // cfg 24 / 25: 3x3 stride-1 convolution on whole-row tiles (128 / 64 pixels x 64 channels) with the activation patch shared
        // by the nine taps (wreg_halo_tile.inc; the record's wgt_frag is the chunk-major fragment pack)
        const bool halo = cfg == SEQ_CFG_HALO128 || cfg == SEQ_CFG_HALO64;
        const int bn = (cfg == 0 || cfg == 3 || cfg == 16 || cfg == 17) ? 256 : ((cfg == 2 || cfg == 9 || cfg == 18) ? 64 : 128);     // cfg 1, 4, 5..8: 128 columns
        const int bm = (cfg == 3 || cfg == 4 || cfg == 9 || cfg == 16) ? 128 : 64;
        const int tilesN = (L.Nst + bn - 1) / bn;
        const int hw = L.Ho * L.Wo;
        const int halo_rpt = halo ? (cfg == SEQ_CFG_HALO128 ? 128 : 64) / L.Wo : 1;
        const int halo_tn = (L.Nst + 63) >> 6;
        const int tiles = fused3 ? L.Ho : fused ? (hw + 31) / 32 : (halo ? ((L.Ho + halo_rpt - 1) / halo_rpt) * halo_tn : ((hw + bm - 1) / bm) * tilesN);
#ifdef SMK_MEASURE
        if (T3 != 0 && fusedp) {
            // pair p = team slots 2p, 2p + 1; 64-row tiles dealt to the pairs (rows per tile evened out when one round covers the
            // image: 961 rows -> 16 tiles of 61); each pair counts its exchanges in bar[8 + p]
            const int npairs = nslots >> 1, pr = slot >> 1, hcu = slot & 1;
            int rt = (hw + npairs - 1) / npairs;
            if (rt > 64) rt = 64;
            const int ptiles = (hw + rt - 1) / rt;
            float *slabs = (float *)((unsigned char *)a.xch + (size_t)((team * SEQ_XCH_PAIRS + pr) * 4) * SEQ_XCH_SLAB);
            for (int img = team; img < a.B && alive; img += 8)
                for (int t = pr; t < ptiles && alive; t += npairs) {
                    unsigned long long *tclk = nullptr;
                    if constexpr (CLK != 0)
                        tclk = (a.clk2 && team == 0 && slot == 0 && img == team && t == pr) ? a.clk2 + SEQ_CLK2_STRIDE * li : nullptr;
                    const TeamWait w{&a, cnt, pending, &ctl[2]};
                    pending = 0;
                    const int fm0 = img * hw + t * rt;
                    int fme = fm0 + rt;
                    if (fme > (img + 1) * hw) fme = (img + 1) * hw;
                    if constexpr (T3 != 0) {
                    if (cfg == SEQ_CFG_C3C1P_L3)
                        alive = c3c1p_tile<256, 1024, 256, CLK>(L, a.L[li + 1], fm0, fme, a.B * hw, hcu, pr, npairs, cnt + 8 + pr, slabs, a, &ctl[2], &ctl[3], smem, tclk, w);
                    else
                        alive = c3c1p_tile<128, 512, 128, CLK>(L, a.L[li + 1], fm0, fme, a.B * hw, hcu, pr, npairs, cnt + 8 + pr, slabs, a, &ctl[2], &ctl[3], smem, tclk, w);
                    }
                }
        } else
#endif
        {
        const int nk = L.Kpad >> 6;
        // K-loop stagger: the workgroups of a team start at K tiles spread over the whole loop (L.kstag)
        const int kt0 = L.kstag ? (slot * nk) / nslots : 0;
        for (int img = team; img < a.B && alive; img += 8)
            for (int t = slot; t < tiles && alive; t += nslots) {
                const int tm = t / tilesN, tn = t - tm * tilesN;
                const int m0 = img * hw + tm * bm, m_end = (img + 1) * hw;
                // (measurement build: the phases of this workgroup's FIRST tile of the layer, team 0 / slot 0)
                unsigned long long *tclk = nullptr;
                if constexpr (CLK != 0)
                    tclk = (a.clk2 && team == 0 && slot == 0 && img == team && t == slot) ? a.clk2 + SEQ_CLK2_STRIDE * li : nullptr;
                const TeamWait w{&a, cnt, pending, &ctl[2]};
                pending = 0;
                if constexpr (T3 != 0)
                if (fused3) {
                    const int fm0 = img * hw + t * L.Wo;
                    if (cfg == SEQ_CFG_C2C3C1_L3) alive = c3c1_tile<256, 1024, 256, CLK, TeamWait, 1>(a.L[li + 1], a.L[li + 2], fm0, fm0 + L.Wo, a.B * hw, slot, nslots, smem, tclk, w, &L, img, t);
                    else alive = c3c1_tile<128, 512, 128, CLK, TeamWait, 1>(a.L[li + 1], a.L[li + 2], fm0, fm0 + L.Wo, a.B * hw, slot, nslots, smem, tclk, w, &L, img, t);
                }
                if (T3 != 0 && fused3) {}
                else if (fused) {
                    const int fm0 = img * hw + t * 32;
                    if (cfg == SEQ_CFG_C3C1_L3) alive = c3c1_tile<256, 1024, 256, CLK>(L, a.L[li + 1], fm0, m_end, a.B * hw, slot, nslots, smem, tclk, w);
                    else alive = c3c1_tile<128, 512, 128, CLK>(L, a.L[li + 1], fm0, m_end, a.B * hw, slot, nslots, smem, tclk, w);
                }
                else if (halo) {
                    const int ty = t / halo_tn, hn0 = (t - ty * halo_tn) * 64;
                    const bool hi = (L.a_stage & SEQ_LDS_HI) != 0;           // a pair's Y image stays in the first 64 KB
                    unsigned char *hsm = smem + (hi ? SEQ_YRES_BYTES : 0);
                    if (cfg == SEQ_CFG_HALO128) alive = wreg_halo_tile<4, NPW, HALO_D128, HALO_SB128, CLK>(L, img, ty, hn0, hsm, tclk, slot, nslots, w, hi);
                    else alive = wreg_halo_tile<2, NPW, HALO_D64, 0, CLK>(L, img, ty, hn0, hsm, tclk, slot, nslots, w, hi);
                }
                else if (cfg == 0) alive = wreg_tile<2, 4, 1, 3, 16, 2, NPW, CLK>(L, 0, m0, m_end, tn * 256, smem, tclk, kt0, w);
                else if (cfg == 1) alive = wreg_tile<2, 2, 2, 3, 16, 2, NPW, CLK>(L, 0, m0, m_end, tn * 128, smem, tclk, kt0, w);
                // 128-row tiles: weight fragments ONE K tile ahead (a k-step is 8 MFMAs here, so the cover in time is that of
                // two tiles at 64 rows; two ahead would need 234 + VGPRs and spill under this kernel's 256)
                else if (cfg == 3) alive = wreg_tile<4, 4, 1, 3, 16, 1, NPW, CLK>(L, 0, m0, m_end, tn * 256, smem, tclk, kt0, w);
                else if (cfg == 4) alive = wreg_tile<4, 2, 2, 3, 16, 1, NPW, CLK>(L, 0, m0, m_end, tn * 128, smem, tclk, kt0, w);
                else if (T3 != 0 && cfg == 9) { if constexpr (T3 != 0) alive = wreg_tile<4, 1, 4, 3, 16, 2, NPW, CLK>(L, 0, m0, m_end, tn * 64, smem, tclk, kt0, w); }     // (128x64: layer1 inside a sequence, smk_tune seq_first_stage)
                // (measurement variant: 64x128 with a 5-deep activation ring and the weight fragments FOUR K tiles ahead)
                else if (T3 != 0 && cfg == 5) { if constexpr (T3 != 0) alive = wreg_tile<2, 2, 2, 5, 16, 4, NPW, CLK>(L, 0, m0, m_end, tn * 128, smem, tclk, kt0, w); }
#ifdef SMK_MEASURE
                // (measurement variants of the 64x128 tile, only in a library built with `make MEASURE=1`; results wrong by construction: 6 = activation tiles never refilled,
                //  7 = weight fragments never refilled, 8 = no MFMA -- which stream sets the K-tile time inside a sequence?)
                else if (cfg == 6) alive = wreg_tile<2, 2, 2, 3, 16 | 0x100, 2, NPW, CLK>(L, 0, m0, m_end, tn * 128, smem, tclk, kt0, w);
                else if (cfg == 7) alive = wreg_tile<2, 2, 2, 3, 16 | 0x200, 2, NPW, CLK>(L, 0, m0, m_end, tn * 128, smem, tclk, kt0, w);
                else if (cfg == 8) alive = wreg_tile<2, 2, 2, 3, 16 | 0x400, 2, NPW, CLK>(L, 0, m0, m_end, tn * 128, smem, tclk, kt0, w);
                // (second set: 10 = MFMA + fragment reads + barriers (no operand refills), 11 = fragment
                //  reads + barriers, 12 = barriers only, 13 = everything but the K-loop barriers, 14 = everything but the fragment reads)
                else if (cfg == 10) alive = wreg_tile<2, 2, 2, 3, 16 | 0x300, 2, NPW, CLK>(L, 0, m0, m_end, tn * 128, smem, tclk, kt0, w);
                else if (cfg == 11) alive = wreg_tile<2, 2, 2, 3, 16 | 0x700, 2, NPW, CLK>(L, 0, m0, m_end, tn * 128, smem, tclk, kt0, w);
                else if (cfg == 12) alive = wreg_tile<2, 2, 2, 3, 16 | 0x1700, 2, NPW, CLK>(L, 0, m0, m_end, tn * 128, smem, tclk, kt0, w);
                else if (cfg == 13) alive = wreg_tile<2, 2, 2, 3, 16 | 0x800, 2, NPW, CLK>(L, 0, m0, m_end, tn * 128, smem, tclk, kt0, w);
                else if (cfg == 14) alive = wreg_tile<2, 2, 2, 3, 16 | 0x1000, 2, NPW, CLK>(L, 0, m0, m_end, tn * 128, smem, tclk, kt0, w);
                // (15..18: the first version's issue order inside a k-step, the A/B arm of cfg 1 / 3 / 0 / 2)
                else if (cfg == 15) alive = wreg_tile<2, 2, 2, 3, 16 | 0x2000, 2, NPW, CLK>(L, 0, m0, m_end, tn * 128, smem, tclk, kt0, w);
                else if (cfg == 16) alive = wreg_tile<4, 4, 1, 3, 16 | 0x2000, 1, NPW, CLK>(L, 0, m0, m_end, tn * 256, smem, tclk, kt0, w);
                else if (cfg == 17) alive = wreg_tile<2, 4, 1, 3, 16 | 0x2000, 2, NPW, CLK>(L, 0, m0, m_end, tn * 256, smem, tclk, kt0, w);
                else if (cfg == 18) alive = wreg_tile<2, 1, 4, 3, 16 | 0x2000, 2, NPW, CLK>(L, 0, m0, m_end, tn * 64, smem, tclk, kt0, w);
#endif
                else alive = wreg_tile<2, 1, 4, 3, 16, 2, NPW, CLK>(L, 0, m0, m_end, tn * 64, smem, tclk, kt0, w);
            }
        }
        if (clk) a.clk[1 + 2 * li] = wall_clock64();
        if (!alive) break;
        if (fused3) {                                              // the triple's second and third record: their time is in the first one's span
            if (clk) a.clk[2 + 2 * li] = a.clk[3 + 2 * li] = a.clk[4 + 2 * li] = a.clk[5 + 2 * li] = wall_clock64();
            li += 2;
        }
        if (fused || fusedp) {                                     // the pair's second record: its time is in the first one's span
            if (clk) a.clk[2 + 2 * li] = a.clk[3 + 2 * li] = wall_clock64();
            ++li;
        }
        if (a.L[li].bar_ord) {                           // (= sync && a layer follows, numbered by launch_conv_seq)
            if (pending) {                               // this workgroup had no tile in the layer: it still has to pass the
                const TeamWait w{&a, cnt, pending, &ctl[2]};       // previous barrier before it may arrive at the next one
                pending = 0;
                if (!w()) break;
            }
            team_arrive(cnt);
            if constexpr (CLK != 0) {                    // (measurement build: when did EVERY slot of team 0 arrive at this barrier?)
                if (a.clk2 && team == 0 && threadIdx.x == 0) a.clk2[SEQ_CLK2_STRIDE * SEQ_MAX + li * 32 + slot] = wall_clock64();
            }
            pending = (unsigned)a.L[li].bar_ord * (unsigned)nslots;
        }
        if (clk) a.clk[2 + 2 * li] = wall_clock64();
    }
    if (!alive || ctl[2]) return;                        // barrier timeout: the host resets the counters (seq_health)
    // the counters return to zero for the next launch: the LAST workgroup of the team to leave resets them
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (prev == (unsigned)nslots - 1) {
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(cnt + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (int i = 8; i < 32; ++i) __hip_atomic_store(cnt + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     // the pairs' exchange counters
            if (a.exit_sem) {
                // pipelined frame step, depth 2: the chip is free for the previous frame's Refine chain + mask head from here on -- the
                // last of the eight teams to leave raises the semaphore its gate polls (instead of a one-thread kernel behind this one)
                const unsigned t = __hip_atomic_fetch_add(a.exit_sem + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t == 7u) {
                    __hip_atomic_store(a.exit_sem + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(a.exit_sem, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

// ---- the fused (conv3 + residual + ReLU, next 1x1) pair as its OWN launch (round 4): batches that do not run the persistent sequence
// (B = 1 .. 4, 9 .. 11, 32, 64, ...) get the pair's benefit -- one launch and one pass over the 1024-channel trunk instead of two --
// without teams: workgroup t owns rows [32 t, 32 t + 32) of the flattened batch (a 1x1 convolution does not care about image
// borders), no barrier, no hoist; the tile routine is the sequence's, bit for bit.
template <int K3, int N3, int N1>
__global__ __launch_bounds__(512, 1) void conv_pair_kernel(const SeqLayer L3, const SeqLayer L1, const int M) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[C3C1Lds<K3, N3, N1>::v];
    const int m0 = (int)blockIdx.x * 32;
    (void)c3c1_tile<K3, N3, N1, 0>(L3, L1, m0, M, M, (int)(blockIdx.x & 31), 32, smem, nullptr, NoHoist());
}

// ... and on 64-row tiles (c3c1s_tile.inc: conv3 in two channel halves, the full Y image in LDS): half the weight bytes per row
template <int K3, int N3, int N1>
__global__ __launch_bounds__(512, 1) void conv_pair64_kernel(const SeqLayer L3, const SeqLayer L1, const int M) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[C3C1SLds<K3, N3, N1>::v];
    c3c1s_tile<K3, N3, N1>(L3, L1, (int)blockIdx.x * 64, M, M, smem);
}

int launch_conv_pair(const SeqLayer &L3, const SeqLayer &L1, int code, int M, void *stream, int rows) {
    if (M < 1 || (code != SEQ_CFG_C3C1_L3 && code != SEQ_CFG_C3C1_L2) || (rows != 32 && rows != 64)) return -1;
    if (rows == 64) {
        const dim3 grid64((M + 63) / 64), block64(512);
        if (code == SEQ_CFG_C3C1_L3) hipLaunchKernelGGL((conv_pair64_kernel<256, 1024, 256>), grid64, block64, 0, (hipStream_t)stream, L3, L1, M);
        else hipLaunchKernelGGL((conv_pair64_kernel<128, 512, 128>), grid64, block64, 0, (hipStream_t)stream, L3, L1, M);
        return hipGetLastError() == hipSuccess ? 0 : -4;
    }
    const dim3 grid((M + 31) / 32), block(512);
    if (code == SEQ_CFG_C3C1_L3) hipLaunchKernelGGL((conv_pair_kernel<256, 1024, 256>), grid, block, 0, (hipStream_t)stream, L3, L1, M);
    else hipLaunchKernelGGL((conv_pair_kernel<128, 512, 128>), grid, block, 0, (hipStream_t)stream, L3, L1, M);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// census: which XCD does block i run on?  (smk_create checks the i % 8 assumption once per context)
__global__ void xcc_census_kernel(int *out) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(xcc & 0xf);
}

int launch_conv_seq(const SeqArgs &a_in, int grid, void *stream) {
    if (a_in.n < 1 || a_in.n > SEQ_MAX || grid < 8 || (grid & 7) || !a_in.bar || !a_in.err) return -1;
    SeqArgs a = a_in;
    unsigned short ord = 0;
    for (int li = 0; li < a.n; ++li) {
        const int cfg = a.L[li].cfg;
        a.L[li].bar_ord = 0;
        const bool first = cfg == SEQ_CFG_C3C1_L3 || cfg == SEQ_CFG_C3C1_L2 || cfg == SEQ_CFG_C3C1P_L3 || cfg == SEQ_CFG_C3C1P_L2;
        if (cfg == SEQ_CFG_C2C3C1_L3 || cfg == SEQ_CFG_C2C3C1_L2) {      // a triple: conv3 (30) and the 1x1 (22) must follow; the 1x1's `sync` counts
            if (li + 2 >= a.n) return -1;
            if (a.L[li + 1].cfg != SEQ_CFG_C2C3C1_MID || a.L[li + 2].cfg != SEQ_CFG_C3C1_2ND) return -1;
            if (a.L[li].Wo + 2 * a.L[li].dil > 35 || a.L[li].Wo > 32) return -1;
            continue;
        }
        if (cfg == SEQ_CFG_C2C3C1_MID) {
            const int pc = li ? a.L[li - 1].cfg : -1;
            if (pc != SEQ_CFG_C2C3C1_L3 && pc != SEQ_CFG_C2C3C1_L2) return -1;
            continue;
        }
        if (first) {                                                     // a pair: the second record must follow, its `sync` counts
            if (li + 1 >= a.n || a.L[li + 1].cfg != SEQ_CFG_C3C1_2ND) return -1;
            if ((cfg == SEQ_CFG_C3C1P_L3 || cfg == SEQ_CFG_C3C1P_L2) && (!a.xch || (grid >> 3) % 2 || (grid >> 4) > SEQ_XCH_PAIRS)) return -1;
            continue;
        }
        if (cfg == SEQ_CFG_C3C1_2ND) {
            const int pc = li ? a.L[li - 1].cfg : -1;
            if (pc != SEQ_CFG_C3C1_L3 && pc != SEQ_CFG_C3C1_L2 && pc != SEQ_CFG_C3C1P_L3 && pc != SEQ_CFG_C3C1P_L2 && pc != SEQ_CFG_C2C3C1_MID) return -1;
        }
        if (a.L[li].sync && li + 1 < a.n) a.L[li].bar_ord = ++ord;
    }
    // the product instantiation carries only the routines the default lists use; triples, pair splits and the deep-ring measurement
    // tile live in the extended one (every routine the kernel carries costs the others registers: profiles/r04w_triples_ab.txt)
    bool triples = false;
    for (int li = 0; li < a.n; ++li) {
        const int cf = a.L[li].cfg;
        triples = triples || cf == SEQ_CFG_C2C3C1_L3 || cf == SEQ_CFG_C2C3C1_L2 || cf == SEQ_CFG_C3C1P_L3 || cf == SEQ_CFG_C3C1P_L2 || cf == 5 || cf == 9;
    }
    if (triples) {
#ifdef SMK_MEASURE
        if (a.clk || a.clk2) hipLaunchKernelGGL((conv_seq_kernel<4, 1, 1>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((conv_seq_kernel<4, 0, 1>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
#else
        // triples, the pair split over two CUs, the deep-ring tile and layer1's 128 x 64 tile each measured a wash or a loss
        // (HISTORY.md 3.1l, 3.1o); the product library does not carry their instantiation (`make MEASURE=1` does)
        return -5;
#endif
    } else if (a.clk || a.clk2)                           // SMK_SEQ_CLK=1 / 2: the build with the stamps (eager runs only)
        hipLaunchKernelGGL((conv_seq_kernel<4, 1>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((conv_seq_kernel<4, 0>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int conv_seq_occupancy() {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_seq_kernel<4, 0>, 512, 0) != hipSuccess) return 0;
    return nb;
}

int xcc_census(int grid, int *out_host) {
    int *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(int) * grid) != hipSuccess) return -4;
    hipLaunchKernelGGL(xcc_census_kernel, dim3(grid), dim3(384), 0, 0, d);
    hipError_t e = hipMemcpy(out_host, d, sizeof(int) * grid, hipMemcpyDeviceToHost);
    hipFree(d);
    return e == hipSuccess ? 0 : -4;
}

}  // namespace smk
