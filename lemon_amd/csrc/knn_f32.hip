// knn_f32.hip -- exact brute-force top-k scan on the fp32 matrix cores of gfx950.
//
// Stands in for faiss IndexFlat{IP,L2}.search as called at run_lemon.py:235-236.
// A workgroup works through SEGMENTS (lemon_plan_segments): a panel of BQ=128 queries against a contiguous range of
// database tiles (BX=128 rows) streamed through LDS; S = Q.X^T is accumulated by v_mfma_f32_32x32x2_f32 in
// ascending k, which is bit-for-bit the float32 fmaf chain of the numeric contract, so the
// scores (and therefore the selected sets, given the index tie rule) are identical to the CPU
// oracle's.  Selection is fused into the tile epilogue: an accumulator element that beats the
// query's current admission threshold (a lower bound of its k-th best score, kept in a VGPR) is appended to the
// owning lane's private half of that query's candidate list (count in a VGPR, list L2-resident: one predicated
// global store, no atomics, no LDS); a list is re-selected by one wavefront when it has grown by ~96 entries.
// The main loop issues every memory instruction by hand (counted waits), the two workgroups of a CU take turns at
// the higher issue priority, and the shares of work are aligned per XCD: DESIGN.md section 4 has the measurements.
//
// Roofline: dense contraction, 2*nq*n*d flop on the fp32 MFMA pipe (157 TFLOP/s dense peak);
// fabric traffic: one query-panel slice + one database tile per unit, shared through the L2 only between
// workgroups that walk the same tiles in step.
#include "knn_common.hpp"
#include "scan_plan.hpp"
#include <algorithm>
#include <vector>

using namespace lemon_knn;
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ---- layout kernels ---------------------------------------------------------------------
// dst[r][8u + (e&1)*4 + (e>>1)] = src[r][8u+e]; zero beyond d.  One thread per (row, 8-group).
__global__ __launch_bounds__(256) void k_permute_rows(const float *__restrict__ src, int64_t n, int d,
                                                      float *__restrict__ dst, int dpad) {
    const int groups = dpad / 8;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * groups) return;
    const int64_t r = t / groups;
    const int u = (int)(t % groups);
    const float *s = src + r * (int64_t)d + 8 * u;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (8 * u + e < d) ? s[e] : 0.0f;
    float4 *o = reinterpret_cast<float4 *>(dst + r * (int64_t)dpad + 8 * u);
    o[0] = make_float4(v[0], v[2], v[4], v[6]);
    o[1] = make_float4(v[1], v[3], v[5], v[7]);
}

// survivors of one 32x32 accumulator tile: a[e] = exact score of (db row jb + (e&3) + 8(e>>2), this
// lane's query); strict '>' against the query's k-th best (rows arrive in ascending index, so an equal
// score with a later index loses).  Appends go to this lane's private half-list.
// (appends address the list as uniform base + 32-bit lane offset and compare row numbers in 32 bits: the 64-bit forms
// cost ~8 more VALU instructions per survivor, and every VALU instruction of the epilogue issues at a fraction of its
// normal rate against the co-resident workgroup's MFMA stream)
template <bool L2, bool UB>
__device__ __forceinline__ void f32_filter_tile(f32x16 a, float th, unsigned jb, float qn,
                                                const float *__restrict__ xnorm, unsigned n, int &ccnt,
                                                char *__restrict__ panel_bytes, unsigned my_off, u64 ub) {
    if (L2) {   // exact key of the numeric contract: -max(0, fma(-2, <q,x>, |q|^2 + |x|^2))
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 xn4 = *reinterpret_cast<const float4 *>(&xnorm[jb + 8 * g]);
            const float xn[4] = {xn4.x, xn4.y, xn4.z, xn4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dd = __builtin_fmaf(-2.0f, a[4 * g + e], qn + xn[e]);
                a[4 * g + e] = -(dd > 0.0f ? dd : 0.0f);
            }
        }
    }
    // Wave-level test first: the group maximum of every lane against its threshold, one ballot.  Then, per
    // half of the group, all eight element ballots are issued back to back (their VALU->SALU latencies
    // overlap) and each element is guarded by a SCALAR branch on its ballot -- the per-element
    // compare -> saveexec -> execz-branch chains of the straightforward form serialised those latencies
    // and cost ~10 % of the launch at N = 40 000, where ~35 rows per wave and tile pass.
    // 16-value maximum in 8 v_max3_f32 (fmaxf() costs 23: hipcc quiets every input with v_max x, x first -- matrix-core
    // results are never signalling, and a quiet NaN element is ignored here exactly as the per-element '>' ignores it)
    const float m = max3(max3(max3(a[0], a[1], a[2]), max3(a[3], a[4], a[5]), max3(a[6], a[7], a[8])),
                         max3(max3(a[9], a[10], a[11]), max3(a[12], a[13], a[14]), a[15]), a[15]);
    if (__ballot(m > th) == 0) return;                   // wave-uniform
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        u64 hb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) hb[i] = __ballot(a[8 * half + i] > th);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (hb[i]) {                                 // scalar branch on a value computed long ago
                const int e = 8 * half + i;
                const unsigned j = jb + (e & 3) + 8 * (e >> 2);
                if (a[e] > th) {                         // (rows >= n: masked to -inf by the caller, last tile only)
                    const u64 key = lemon_make_key(a[e], j);
                    if (!UB || key < ub) {
                        *reinterpret_cast<u64 *>(panel_bytes + (my_off + 8u * append_slot(ccnt))) = key;
                        ++ccnt;
                    }
                }
            }
        }
    }
}

// One workgroup = 128 queries x a range of 128-row database tiles.  Wave w owns queries 32w..32w+31
// (B operand, one query per lane pair) against all 128 rows of the tile (A operand, 4 row tiles):
// acc_i[e] = <x_(32i + (e&3) + 8(e>>2) + 4h), q_(32w + lane&31)> accumulated in ascending k.
template <bool L2, bool PROF, bool UB = false>
__global__ __launch_bounds__(NT, 2) void k_scan_f32(ScanParams p) {
    // diagnostic instantiation: per-phase cycle sums of wave 0 (loop incl. barriers, filter, maintenance, final)
    unsigned long long ts = 0, ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0;
#define PH_STAMP(acc) do { if (PROF) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - ts; ts = now_; } } while (0)
    unsigned long long clk0 = 0, rt0 = 0;
    if (PROF) { ts = __builtin_amdgcn_s_memtime(); clk0 = ts; rt0 = __builtin_amdgcn_s_memrealtime(); }
    __shared__ __attribute__((aligned(16))) float s_tile[2][2][BQ * BK];  // [buf][Q|X][row*32+..] 64 KiB
    __shared__ __attribute__((aligned(16))) u64 s_keys[NT / 64][256];    // rank-merge scratch, one per wave
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;

    // work of this workgroup: a run of SEGMENTS = (panel, first tile, tiles, piece number inside the panel).  With a plan
    // (lemon_plan_segments: equal unit counts per workgroup, XCD-aware, see there) they come from a table; without one
    // (LEMON_SPLITS) the workgroup is one (panel, split) rectangle.
    int seg = 0, seg_end = 1;
    if (p.plan) { seg = p.plan[blockIdx.x]; seg_end = p.plan[blockIdx.x + 1]; }
    // Fair shares of the matrix pipe.  The two wavefronts of a SIMD (one from each co-resident workgroup) both have MFMAs
    // ready most of the time, and the issue arbiter prefers the OLDER one: measured at 131 072^2 x 512, the workgroups
    // in the even wave slots lived 106 ms, those in the odd slots 130 ms -- for the last fifth of the launch half the
    // waves ran alone, at little more than half the pipe's rate.  So the two take turns: issue priority 1 for the wave
    // whose slot parity equals bit 13 of the chip-wide 100 MHz counter (82 us per turn, re-evaluated after every tile),
    // 0 for the other.  (The epilogue raises itself to 3 either way.)
    const unsigned slot_parity = __builtin_amdgcn_s_getreg((1 << 11) | (0 << 6) | 4) & 1u;   // HW_ID[0]: wave slot on this SIMD
#define F32_BASE_PRIO()                                                                                \
    do {                                                                                               \
        if (p.fair >= 100) {       /* diagnostic: static priority by wave slot (100: even slots high, 101: odd) */ \
            if ((slot_parity ^ (unsigned)p.fair ^ 1u) & 1u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); \
        } else if (p.fair) {                                                                           \
            if ((((unsigned)__builtin_amdgcn_s_memrealtime() >> p.fair) ^ slot_parity) & 1u) __builtin_amdgcn_s_setprio(1); \
            else __builtin_amdgcn_s_setprio(0);                                                        \
        } else __builtin_amdgcn_s_setprio(0);                                                          \
    } while (0)
    F32_BASE_PRIO();
    const int KT = p.dpad / BK;
    const int dpad = p.dpad;
    const unsigned voff = (unsigned)(((tid >> 3) * dpad + 4 * (tid & 7)) * 4);   // row tid>>3 (+ 32 i), chunk tid&7: bytes
    const unsigned voff1 = voff + (unsigned)(NT / 8) * dpad * 4, voff2 = voff1 + (unsigned)(NT / 8) * dpad * 4,
                   voff3 = voff2 + (unsigned)(NT / 8) * dpad * 4;
    u64 *cand_panel = p.cand + (int64_t)blockIdx.x * BQ * PAIR_CAP;
    const int qrow_l = 32 * wave + l31;
    char *panel_bytes = reinterpret_cast<char *>(cand_panel);
    const unsigned my_off = (unsigned)(qrow_l * PAIR_CAP + h * (PAIR_CAP / 2)) * 8u;   // this lane's half-list, in bytes

    for (; seg < seg_end; ++seg) {                      // workgroup-uniform
    int panel, t_begin, ntile, split;
    if (p.plan) {
        const int4 sg = reinterpret_cast<const int4 *>(p.plan + p.plan_segs)[seg];
        panel = sg.x; t_begin = sg.y; ntile = sg.z; split = sg.w;
    } else {
        panel = blockIdx.x / p.splits; split = blockIdx.x % p.splits;
        t_begin = split * p.tiles_per_split;
        ntile = p.n_tiles - t_begin < p.tiles_per_split ? p.n_tiles - t_begin : p.tiles_per_split;
    }
    panel = __builtin_amdgcn_readfirstlane(panel); t_begin = __builtin_amdgcn_readfirstlane(t_begin);
    ntile = __builtin_amdgcn_readfirstlane(ntile); split = __builtin_amdgcn_readfirstlane(split);
    const int64_t q0 = (int64_t)panel * BQ;
    const int total = ntile * KT;

    // lane-private candidate state (see knn_common.hpp "pair lists")
    const bool qvalid = q0 + qrow_l < p.nq;
    const float my_qn = L2 ? p.qnorm[q0 + qrow_l] : 0.0f;
    const u64 my_ub = UB ? p.ub[q0 + qrow_l] : ~0ull;
    int ccnt = 0, clast = 0;
    float th = qvalid ? -INFINITY : INFINITY;    // admission bound: max(own bound, bound imported from the other pieces)
    // Shared bounds.  A query panel is scanned by 2-3 workgroups (stream-K pieces) or by `splits` of them, each over its
    // own rows.  The k-th best score inside ANY subset of the rows is a lower bound of the k-th best over all of them, so
    // a piece may drop every row that is strictly below another piece's bound.  Each workgroup publishes its own bound
    // (order-encoded, atomicMax) after a re-selection and reads the panel's maximum once per tile; it uses the next lower
    // float of what it reads, because a row that TIES the imported bound may still win on the index.  The pieces' cold
    // starts then cost one warm-up in total instead of one each: at the headline shape ~35 % fewer admissions.
    float th_own = th;                           // bound derived from this workgroup's own list (what gets published)
    unsigned pub = 0;                            // last value read from th_pub (0 = nothing published yet)
    // (hand-issued like the operand loads, device-coherent: hipcc does not see those, so its own wait for this value would
    // be vmcnt(0) = drain the operand prefetch once per tile.  At least 16 operand loads follow every fetch before its use.)
#define F32_PUB_FETCH() do { if (p.th_pub) asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 sc0 sc1" : "=v"(pub) : "v"(4u * (unsigned)qrow_l), "s"(p.th_pub + q0) : "memory"); } while (0)
    F32_PUB_FETCH();

    // Operand staging global -> registers -> LDS with TWO register sets: set (s & 1) carries stage s.
    // At step s the stage s+1 is committed to the other LDS buffer and the freed set is re-issued for
    // stage s+3, so every load has two full k-steps (128 MFMAs per wave) to land.  (With one set the
    // distance was one step and the wait for it cost ~18 % of the launch under the fabric load of 512
    // workgroups re-streaming their query panels.)  KT is even (dpad is a multiple of 64), so tiles
    // start on even stages and the register sets have static names.
    f32x4 ra_q0, ra_q1, ra_q2, ra_q3, ra_x0, ra_x1, ra_x2, ra_x3;
    f32x4 rb_q0, rb_q1, rb_q2, rb_q3, rb_x0, rb_x1, rb_x2, rb_x3;
    const float *qbase = p.qp + q0 * dpad;
    const float *xbase = p.xp + (int64_t)t_begin * BX * dpad;
    int lk = 0, lj = 0, ls = 0;                  // load cursor: next stage to issue = ls = lj * KT + lk
    // The operand loads are issued by hand: `global_load_dwordx4 v, v_offset32, s[base]` with the wave-uniform row-block
    // base in SGPRs, and ONE counted wait in front of a stage's LDS writes.  Left to hipcc each load costs a 64-bit VALU
    // address add and each LDS write its own s_waitcnt; a stand-alone replica of this loop (tools/micro/scan_loop.hip)
    // runs at 0.934 of the fp32 MFMA peak this way and at 0.894 the other.  Always eight loads per stage (past the end
    // of the segment the first stage is fetched again and never consumed), so that the count is static:
    // vmcnt(8) = "everything but the eight newest vector-memory operations has completed" = the other register set's
    // loads (or, right after an epilogue, its appends) may still be in flight, this set's cannot.
    // (s_nop 4: a VMEM instruction that reads an SGPR written by the SALU less than 5 wait states earlier sees the OLD
    // value -- hipcc pads its own loads and does not look inside inline asm; all eight bases are inputs of ONE statement)
#define F32_ISSUE(S)                                                                                   \
    do {                                                                                               \
        const bool live_ = ls < total;                                                                 \
        const float *qs_ = live_ ? qbase + lk * BK : qbase;                                            \
        const float *xs_ = live_ ? xbase + (int64_t)lj * BX * dpad + lk * BK : xbase;                  \
        asm volatile("s_nop 4\n\t"                                                                     \
                     "global_load_dwordx4 %0, %8, %12\n\tglobal_load_dwordx4 %1, %9, %12\n\t"    \
                     "global_load_dwordx4 %2, %10, %12\n\tglobal_load_dwordx4 %3, %11, %12\n\t"  \
                     "global_load_dwordx4 %4, %8, %13\n\tglobal_load_dwordx4 %5, %9, %13\n\t"          \
                     "global_load_dwordx4 %6, %10, %13\n\tglobal_load_dwordx4 %7, %11, %13"            \
                     : "=&v"(r##S##_q0), "=&v"(r##S##_q1), "=&v"(r##S##_q2), "=&v"(r##S##_q3),         \
                       "=&v"(r##S##_x0), "=&v"(r##S##_x1), "=&v"(r##S##_x2), "=&v"(r##S##_x3)          \
                     : "v"(voff), "v"(voff1), "v"(voff2), "v"(voff3), "s"(qs_), "s"(xs_) : "memory");  \
        ++ls; if (++lk == KT) { lk = 0; ++lj; }                                                        \
    } while (0)
#define F32_LANDED(S, N)                                                                               \
    asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(r##S##_q0), "+v"(r##S##_q1), "+v"(r##S##_q2), "+v"(r##S##_q3), \
                                             "+v"(r##S##_x0), "+v"(r##S##_x1), "+v"(r##S##_x2), "+v"(r##S##_x3))
#define F32_ST1(V, OFF) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(wa), "v"(V), "n"(OFF) : "memory")
#define F32_COMMIT(S, BUF)                                                                             \
    do {                                                                                               \
        F32_LANDED(S, 8);                                                                              \
        F32_ST1(r##S##_q0, (BUF) * 32768); F32_ST1(r##S##_q1, (BUF) * 32768 + 4096);                   \
        F32_ST1(r##S##_q2, (BUF) * 32768 + 8192); F32_ST1(r##S##_q3, (BUF) * 32768 + 12288);           \
        F32_ST1(r##S##_x0, (BUF) * 32768 + 16384); F32_ST1(r##S##_x1, (BUF) * 32768 + 20480);          \
        F32_ST1(r##S##_x2, (BUF) * 32768 + 24576); F32_ST1(r##S##_x3, (BUF) * 32768 + 28672);          \
    } while (0)
#define F32_FRAG(F, BUF, U)                                                                            \
    asm volatile("ds_read_b128 %0, %5 offset:%7\n\tds_read_b128 %1, %6 offset:%8\n\tds_read_b128 %2, %6 offset:%9\n\t" \
                 "ds_read_b128 %3, %6 offset:%10\n\tds_read_b128 %4, %6 offset:%11"                     \
                 : "=&v"(F##_b), "=&v"(F##_0), "=&v"(F##_1), "=&v"(F##_2), "=&v"(F##_3)                \
                 : "v"(aq[U]), "v"(ax[U]), "n"((BUF) * 32768), "n"((BUF) * 32768 + 16384),             \
                   "n"((BUF) * 32768 + 20480), "n"((BUF) * 32768 + 24576), "n"((BUF) * 32768 + 28672) : "memory")
#define F32_WAIT(F, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(F##_b), "+v"(F##_0), "+v"(F##_1), "+v"(F##_2), "+v"(F##_3))
#define F32_MFMA_F(F, INIT)                                                                            \
    do {                                                                                               \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                                \
            if ((INIT) && m == 0) {                                                                    \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_0[m], F##_b[m], zero16, 0, 0, 0);      \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_1[m], F##_b[m], zero16, 0, 0, 0);      \
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_2[m], F##_b[m], zero16, 0, 0, 0);      \
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_3[m], F##_b[m], zero16, 0, 0, 0);      \
                continue;                                                                              \
            }                                                                                          \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_0[m], F##_b[m], acc0, 0, 0, 0);            \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_1[m], F##_b[m], acc1, 0, 0, 0);            \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_2[m], F##_b[m], acc2, 0, 0, 0);            \
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(F##_3[m], F##_b[m], acc3, 0, 0, 0);            \
        }                                                                                              \
    } while (0)
#define F32_STEP(BUF, SET, it_, FIRST)                                                                 \
    do {                                                                                               \
        F32_FRAG(fb, BUF, 1); F32_WAIT(fa, 5);                                                         \
        if (FIRST) F32_MFMA_F(fa, true); else F32_MFMA_F(fa, false);                                   \
        F32_FRAG(fa, BUF, 2); F32_WAIT(fb, 5); F32_MFMA_F(fb, false);                                  \
        if (!abl_ld) {                                                                                 \
            F32_COMMIT(SET, (BUF) ^ 1);                                                                \
            F32_ISSUE(SET);                                                                            \
            F32_FRAG(fb, BUF, 3); F32_WAIT(fa, 13);                                                    \
        } else {                                                                                       \
            F32_FRAG(fb, BUF, 3); F32_WAIT(fa, 5);                                                     \
        }                                                                                              \
        F32_MFMA_F(fa, false);                                                                         \
        if (!abl_bar) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : : "memory"); \
        F32_FRAG(fa, (BUF) ^ 1, 0); F32_WAIT(fb, 5); F32_MFMA_F(fb, false);                            \
    } while (0)

    f32x4 fa_b, fa_0, fa_1, fa_2, fa_3, fb_b, fb_0, fb_1, fb_2, fb_3;   // two fragment sets
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float *)&s_tile[0][0][0];
    const unsigned wa = lds0 + 4u * (unsigned)swz(tid >> 3, tid & 7);
    unsigned aq[4], ax[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        aq[u] = lds0 + 4u * (unsigned)swz(qrow_l, 2 * u + h);
        ax[u] = lds0 + 4u * (unsigned)swz(l31, 2 * u + h);
    }

    f32x16 acc0, acc1, acc2, acc3, zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.0f;
    acc0 = zero16; acc1 = zero16; acc2 = zero16; acc3 = zero16;

    const bool abl_ld = PROF && (p.ablate & 1), abl_bar = PROF && (p.ablate & 2);
    F32_ISSUE(a);                                // stage 0
    F32_ISSUE(b);                                // stage 1
    F32_COMMIT(a, 0);
    F32_ISSUE(a);                                // stage 2
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    F32_FRAG(fa, 0, 0);

    int jl = 0;                                  // tile of the stage pair being computed
    for (int it = 0; it < total; it += 2) {
        F32_STEP(0, b, it, (it % KT) == 0);      // even stage: LDS buffer 0; set b holds stage it+1
        F32_STEP(1, a, it + 1, false);           // odd stage: LDS buffer 1; set a holds stage it+2
        const bool tile_done = ((it + 2) % KT) == 0;

        if (tile_done) {
            // ---- epilogue: filter into the lane-private half-lists, then zero the accumulators ----
            // (raised issue priority: the other three waves of the workgroup wait at the barrier for this
            // one, while the co-resident workgroup's waves keep the MFMA pipe busy either way)
            PH_STAMP(ph0);
            __builtin_amdgcn_s_setprio(3);
            const unsigned jb = (unsigned)(t_begin + jl) * BX + 4 * h;
            const int ccnt_in = ccnt;
            if (p.th_pub) {
                asm volatile("s_waitcnt vmcnt(16)" : "+v"(pub));
                if (pub > 1u) th = fmaxf(th, lemon_ord2f(pub - 1u));   // strictly below the published bound
            }
            if ((unsigned)(t_begin + jl + 1) * BX > (unsigned)p.n) {   // last tile of the database (uniform): padding rows never pass
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned j = jb + (e & 3) + 8 * (e >> 2);
                    if (j >= (unsigned)p.n) acc0[e] = -INFINITY;
                    if (j + 32 >= (unsigned)p.n) acc1[e] = -INFINITY;
                    if (j + 64 >= (unsigned)p.n) acc2[e] = -INFINITY;
                    if (j + 96 >= (unsigned)p.n) acc3[e] = -INFINITY;
                }
            }
            if (PROF && (p.ablate & 16)) {           // diagnostic: bit 4 = time the bare accumulator read-out (64-value max)
                float mx = acc0[0];
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(fmaxf(mx, acc0[e]), fmaxf(fmaxf(acc1[e], acc2[e]), acc3[e]));
                if (mx == 123.456f) ph3 += 1;       // keep it alive
                PH_STAMP(ph3);                       // reported in the 'final' column
            }
            if (PROF && (p.ablate & (32 | 128)) && jl > 4) {
                // diagnostic: what does the co-resident MFMA stream cost an epilogue per instruction CLASS?  The filter is
                // replaced by instructions of one kind: bit 5 = 512 VALU (4 independent chains; measured 12.7 cycles each
                // instead of 4), bit 7 = 256 ballot + scalar-branch pairs (never taken)
                float x0 = acc0[0], x1 = acc1[0], x2 = acc2[0], x3 = acc3[0];
                                if (p.ablate & 32) {
#pragma unroll 1
                    for (int r = 0; r < 8; ++r)
                        asm volatile(".rept 16\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4\n\t.endr"
                                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(th));
                } else {
#pragma unroll 1
                    for (int r = 0; r < 256; ++r) {
                        if (__ballot(x0 > 3.0e38f)) x1 += 1.0f;     // v_cmp -> SGPR -> scalar branch, never taken
                        asm volatile("" : "+v"(x0));
                    }
                }
                if (x0 + x1 + x2 + x3 == 1.2345f) ph3 += 1;    // keep alive
            } else
            if (!(PROF && (p.ablate & 4) && jl > 4)) {     // diagnostic: bit 2 = skip the filter after 5 tiles
            f32_filter_tile<L2, UB>(acc0, th, jb, my_qn, p.xnorm, (unsigned)p.n, ccnt, panel_bytes, my_off, my_ub);
            f32_filter_tile<L2, UB>(acc1, th, jb + 32, my_qn, p.xnorm, (unsigned)p.n, ccnt, panel_bytes, my_off, my_ub);
            f32_filter_tile<L2, UB>(acc2, th, jb + 64, my_qn, p.xnorm, (unsigned)p.n, ccnt, panel_bytes, my_off, my_ub);
            f32_filter_tile<L2, UB>(acc3, th, jb + 96, my_qn, p.xnorm, (unsigned)p.n, ccnt, panel_bytes, my_off, my_ub);
            }
            if (PROF && (p.ablate & 8) && jl > 4) ccnt = ccnt_in;   // diagnostic: bit 3 = appends land but are forgotten
            ++jl;                                    // (the accumulators are not zeroed: the next tile's first MFMAs take C = 0)

            // ---- maintenance: select the exact top-kk of queries whose lists grew enough ----
            PH_STAMP(ph1);
            const int pair = ccnt + __shfl_xor(ccnt, 32);
            const bool warm = th == -INFINITY && pair >= p.kk;   // no bound at all yet, own or imported
            const bool stale = pair >= p.kk && pair - clast >= p.stale;
            const bool full = ccnt > PAIR_CAP / 2 - BX / 2;
            u64 todo = __ballot(qvalid && (warm || stale || full));
            todo = (todo | (todo >> 32)) & 0xffffffffull;          // one bit per query of this wave
            if (todo) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's appends are visible
                do {
                    const int r = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    u64 *list = cand_panel + (int64_t)(32 * wave + r) * PAIR_CAP;
                    const int n0 = __builtin_amdgcn_readlane(ccnt, r), n1 = __builtin_amdgcn_readlane(ccnt, r + 32);
                    int kept;
                    const u64 kth = pair_select_loose(list, n0, n1, p.kk, lane, &kept);
                    if (l31 == r) {                    // both lanes of the pair take the new state
                        ccnt = h == 0 ? kept : 0;
                        clast = kept;
                        th_own = lemon_key_score(kth); // <= the exact kk-th best: a valid admission threshold
                        th = fmaxf(th, th_own);
                        if (p.th_pub && h == 0) atomicMax(&p.th_pub[q0 + 32 * wave + r], lemon_f2ord(th_own));
                    }
                } while (todo);
            }
            // the bound the other pieces of this panel have published by now, for the NEXT tile's epilogue: fetched here, a
            // whole tile ahead of its use and in front of >= 16 operand loads (fetched at the top of the tile it was the
            // NEWEST load at its use, and the wait for it, vmcnt(0), drained the operand prefetch once per tile)
            F32_PUB_FETCH();
            F32_BASE_PRIO();
            PH_STAMP(ph2);
        }
    }
    F32_WAIT(fa, 0);
#undef F32_STEP
#undef F32_MFMA_F
#undef F32_WAIT
#undef F32_FRAG
#undef F32_ST1
#undef F32_COMMIT
#undef F32_ISSUE

    // ---- final: sort every query's best kk, write the result rows ------------------------------
    PH_STAMP(ph0);
    F32_LANDED(a, 0); F32_LANDED(b, 0);                 // the loads issued past the end of the segment, and the appends
    asm volatile("" : "+v"(pub));
#undef F32_LANDED
#undef F32_PUB_FETCH
    for (int r = 0; r < 32; ++r) {
        const int row = 32 * wave + r;
        const int64_t q = q0 + row;
        if (q >= p.nq) break;
        u64 *list = cand_panel + (int64_t)row * PAIR_CAP;
        const int n0 = __builtin_amdgcn_readlane(ccnt, r), n1 = __builtin_amdgcn_readlane(ccnt, r + 32);
        int pos;
        // (a piece of a panel only has to deliver its best kk: k_merge rank-selects the pieces' union, sorted or not)
        const u64 key = p.splits > 1 ? pair_final_topk<false>(list, n0, n1, p.kk, lane, s_keys[wave], &pos)
                                     : pair_final_topk<true>(list, n0, n1, p.kk, lane, s_keys[wave], &pos);
        write_out_row(p, split, q, pos, key);
    }
    PH_STAMP(ph3);
    __syncthreads();                                    // LDS tiles and lists are reused by the next segment
    }
    if (PROF && tid == 0) {
        atomicAdd(&p.phase_dbg[0], ph0); atomicAdd(&p.phase_dbg[1], ph1);
        atomicAdd(&p.phase_dbg[2], ph2); atomicAdd(&p.phase_dbg[3], ph3);
        const unsigned long long rt = __builtin_amdgcn_s_memrealtime() - rt0;   // this workgroup's life, 100 MHz ticks
        if (blockIdx.x == 0) {
            p.phase_dbg[4] = __builtin_amdgcn_s_memtime() - clk0;
            p.phase_dbg[5] = rt;
        }
        // spread of the workgroups' lives, overall and by the parity of the wave slot wave 0 sits in (HW_ID[3:0]): the two
        // workgroups of a CU share the matrix pipe, and an arbiter that prefers one of them shows up here
        const unsigned slot = slot_parity;
        atomicMax(&p.phase_dbg[6], rt);
        atomicMin(&p.phase_dbg[7], rt);
        atomicAdd(&p.phase_dbg[8 + 2 * slot], rt);
        atomicAdd(&p.phase_dbg[9 + 2 * slot], 1ull);
        if (blockIdx.x < 4096) p.phase_dbg[16 + blockIdx.x] = rt;
    }
#undef PH_STAMP
#undef F32_BASE_PRIO
}

// merge the per-split sorted lists of one query (one wavefront per query)
__global__ __launch_bounds__(256) void k_merge(const u64 *__restrict__ part, int splits, const int *__restrict__ pieces,
                                               int64_t nq_pad, int64_t nq, int kk, int metric,
                                               float *__restrict__ D, int64_t *__restrict__ I) {
    __shared__ __attribute__((aligned(16))) u64 s_keys[4][256];
    __shared__ __attribute__((aligned(16))) u64 s_best[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= nq) return;
    u64 best = 0;  // lane i: i-th best so far
    if (pieces) splits = pieces[q / BQ];      // planned decomposition: this panel's number of pieces
    const int total = splits * kk;
    for (int base = 0; base < total; base += 192) {
        u64 v[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int e = base + 64 * t + lane;
            u64 key = 0;
            if (e < total) {
                const int s = e / kk, pos = e % kk;
                key = part[((int64_t)s * nq_pad + q) * kk + pos];
            }
            v[t] = key;
        }
        const Ranked r = wave_rank_keys(best, v[0], v[1], v[2], 256, s_keys[wave], lane);
        s_best[wave][lane] = 0;
        __builtin_amdgcn_wave_barrier();
        if (best && r.r0 < kk) s_best[wave][r.r0] = best;
        if (v[0] && r.r1 < kk) s_best[wave][r.r1] = v[0];
        if (v[1] && r.r2 < kk) s_best[wave][r.r2] = v[1];
        if (v[2] && r.r3 < kk) s_best[wave][r.r3] = v[2];
        __builtin_amdgcn_wave_barrier();
        best = s_best[wave][lane];
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < kk) {
        float dv; int64_t iv;
        if (best) {
            const float s = lemon_key_score(best);
            dv = (metric == LEMON_METRIC_L2) ? -s : s;
            iv = (int64_t)lemon_key_index(best);
        } else {
            dv = (metric == LEMON_METRIC_L2) ? FLT_MAX : -FLT_MAX;
            iv = -1;
        }
        D[q * kk + lane] = dv;
        I[q * kk + lane] = iv;
    }
}

__global__ void k_fill_empty(float *D, int64_t *I, int64_t total, int metric) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) { D[i] = (metric == LEMON_METRIC_L2) ? FLT_MAX : -FLT_MAX; I[i] = -1; }
}

}  // namespace

// -----------------------------------------------------------------------------------------
// host side
// -----------------------------------------------------------------------------------------
int lemon_permute_rows(const float *src, int64_t n, int d, float *dst, int dpad, hipStream_t s) {
    if (n <= 0) return LEMON_OK;
    const int64_t threads = n * (dpad / 8);
    hipLaunchKernelGGL(k_permute_rows, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, src, n, d,
                       dst, dpad);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

int lemon_launch_merge(const u64 *part, int splits, const int *pieces, int64_t nq_pad, int64_t nq,
                       int kk, int metric, float *D, int64_t *I, hipStream_t stream) {
    hipLaunchKernelGGL(k_merge, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, stream, part, splits, pieces,
                       nq_pad, nq, kk, metric, D, I);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

int lemon_parse_ablate(const char *kernel, int allowed, int *out) {
    *out = 0;
    const char *e = getenv("LEMON_ABLATE");
    if (!e || !*e) return LEMON_OK;
    char *end = nullptr;
    const long v = strtol(e, &end, 0);
    if (end == e || *end != '\0' || v < 0 || (v & ~(long)allowed)) {
        lemon_set_error("LEMON_ABLATE=%s: %s accepts combinations of the bits 0x%x only", e, kernel, allowed);
        return LEMON_E_INVALID;
    }
    *out = (int)v;
    return LEMON_OK;
}

int lemon_fill_empty(float *D, int64_t *I, int64_t total, int metric, hipStream_t stream) {
    if (total <= 0) return LEMON_OK;
    hipLaunchKernelGGL(k_fill_empty, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, D, I, total, metric);
    LEMON_HIP_CHECK(hipGetLastError());
    return LEMON_OK;
}

// enough workgroups to fill 256 CUs x 2 resident 1.5 times; every split keeps >= 8 tiles so the
// per-split warm-up (first-tile selection, final sort, merge) amortises
void lemon_plan_splits(int panels, int n_tiles, int *splits_out, int *tiles_per_split_out) {
    int splits = 1;
    static const int forced = [] { const char *e = getenv("LEMON_SPLITS"); return e ? atoi(e) : 0; }();   // tuning knob
    if (forced > 0) {
        splits = forced;
        if (splits > n_tiles) splits = n_tiles;
    } else if (panels < 768) {
        splits = (768 + panels - 1) / panels;
        int max_splits = n_tiles / 8;
        if (max_splits < 1) max_splits = 1;
        if (splits > max_splits) splits = max_splits;
    }
    const int tiles_per_split = (n_tiles + splits - 1) / splits;
    *splits_out = (n_tiles + tiles_per_split - 1) / tiles_per_split;
    *tiles_per_split_out = tiles_per_split;
}

// Planned ("stream-K", XCD-aware) decomposition of the fp32 scan.
//
// Balance.  One workgroup alone on a CU reaches well under the MFMA rate of two co-resident ones, so a grid that is not
// a multiple of 2 x CUs pays a whole extra round (391 panels ran as slowly as 512).  Instead the panels x tiles unit
// space is cut into equal shares, one per resident slot; a share that touches several panels yields one partial list per
// panel touched and k_merge combines the pieces of a panel.
//
// Locality.  A workgroup re-streams its query panel (128 x d x 4 B) and streams a database tile of the same size per
// unit; the 64 workgroups of an XCD move 32 MB per unit time through a 4 MB L2, so NOTHING hits unless workgroups read
// the same tile at the same time.  With plain contiguous shares (round 1) the shares start at arbitrary tiles: at the
// headline shape PMC FETCH_SIZE was 2.0 x the scan-model bytes and the launch ran at 0.80 of the MFMA peak, against
// 0.89 at 262 144^2 where every share happens to be four whole panels walked from tile 0 in step.  So the shares are
// built per XCD (workgroup b runs on XCD b % 8: round-robin dispatch; panel p belongs to XCD p % 8) and ALIGNED: with
// u units per workgroup and T tiles per panel, T >= u: as many workgroups as the XCD has panels take tiles [0, u) of
// one panel each -- they walk the same tiles in step -- and the other workgroups share the tails [u, T); T < u: every
// workgroup first takes floor(u / T) whole panels (again from tile 0, in step), the rest is shared out.  A segment is
// (panel, first tile, tiles, piece number in the panel); workgroup b owns segments plan[b] .. plan[b+1].
// (the planner itself is plain C++ in scan_plan.hpp: it is also built with gcc under ASan / UBSan and fuzzed on the CPU)
static void lemon_plan_segments(int panels, int n_tiles, LemonPlan &plan) {
    static const int slots = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return 2 * (cus > 0 ? cus : 256);
    }();
    static const int xcds = [] { const char *e = getenv("LEMON_XCDS"); return e && atoi(e) > 0 ? atoi(e) : 8; }();   // 1 = not XCD-aware
    static const int seg_cost = [] { const char *e = getenv("LEMON_SEG_COST"); return e ? atoi(e) : 3; }();   // tile times per segment
    lemon_plan_segments_host(panels, n_tiles, slots, xcds, seg_cost, plan);
}

extern "C" int lemon_debug_scan_plan(int panels, int n_tiles, int *grid, int *splits, int *seg_begin, int cap_wgs, int *pieces,
                                     int *segs, int cap_segs, int *n_segs) {
    LEMON_REQUIRE(panels > 0 && n_tiles > 0, "panels, n_tiles > 0");
    LEMON_REQUIRE(grid && splits && seg_begin && pieces && segs && n_segs, "output pointers");
    LemonPlan plan;
    lemon_plan_segments(panels, n_tiles, plan);
    LEMON_REQUIRE(plan.grid <= cap_wgs && (int)(plan.segs.size() / 4) <= cap_segs, "plan fits the output buffers");
    *grid = plan.grid; *splits = plan.splits; *n_segs = (int)(plan.segs.size() / 4);
    std::copy(plan.seg_begin.begin(), plan.seg_begin.end(), seg_begin);
    std::copy(plan.pieces.begin(), plan.pieces.end(), pieces);
    std::copy(plan.segs.begin(), plan.segs.end(), segs);
    return LEMON_OK;
}

// the plan of (panels, n_tiles) on the device: [grid + 1] segment offsets | [panels] pieces | pad to 4 ints | segments.
// Four recent shapes are kept (least recently used slot replaced); the upload is ordered on the stream, so a launch
// queued earlier that still reads the slot's old content finishes first, and the source vector lives in the slot.
static int lemon_get_plan(lemon_index_t *idx, int panels, int n_tiles, hipStream_t stream, const lemon_index::PlanSlot **out) {
    lemon_index::PlanSlot *hit = nullptr, *lru = &idx->plan_slots[0];
    for (auto &sl : idx->plan_slots) {
        if (sl.dev && sl.panels == panels && sl.tiles == n_tiles) hit = &sl;
        if (sl.stamp < lru->stamp) lru = &sl;
    }
    if (!hit) {
        LemonPlan plan;
        lemon_plan_segments(panels, n_tiles, plan);
        if (!lru->host) lru->host = new std::vector<int>();
        else LEMON_HIP_CHECK(hipStreamSynchronize(stream));            // the previous upload from this vector has completed
        std::vector<int> &buf = *lru->host;
        buf.assign(plan.seg_begin.begin(), plan.seg_begin.end());
        const int po = (int)buf.size();
        buf.insert(buf.end(), plan.pieces.begin(), plan.pieces.end());
        while (buf.size() % 4) buf.push_back(0);
        const int so = (int)buf.size();
        buf.insert(buf.end(), plan.segs.begin(), plan.segs.end());
        if ((int64_t)buf.size() > lru->ints) {
            LEMON_HIP_CHECK(hipStreamSynchronize(stream));             // an earlier launch may still read the old buffer
            if (lru->dev) (void)hipFree(lru->dev);
            lru->dev = nullptr; lru->ints = 0;
            if (hipMalloc((void **)&lru->dev, buf.size() * sizeof(int)) != hipSuccess) {
                lemon_set_error("scan plan allocation failed");
                return LEMON_E_NOMEM;
            }
            lru->ints = (int64_t)buf.size();
        }
        LEMON_HIP_CHECK(hipMemcpyAsync(lru->dev, buf.data(), buf.size() * sizeof(int), hipMemcpyHostToDevice, stream));
        lru->panels = panels; lru->tiles = n_tiles; lru->grid = plan.grid; lru->splits = plan.splits;
        lru->pieces_off = po; lru->segs_off = so;
        hit = lru;
    }
    hit->stamp = ++idx->plan_clock;
    *out = hit;
    return LEMON_OK;
}

// qp_row_bytes: bytes of one staged query row (dpad*4 for the fp32 scan, dpad_h*2 for the bf16 one)
int lemon_ensure_search_ws(lemon_index_t *idx, int64_t nq_pad, int splits, int64_t n_wg, int qp_row_bytes,
                           int cand_cap, hipStream_t stream) {
    const int64_t part_elems = (splits > 1) ? (int64_t)splits * nq_pad * LEMON_MAX_K : 0;
    const int64_t cand_rows = n_wg * BQ * (cand_cap / CAP);         // in units of CAP-entry rows
    int64_t row_bytes = (int64_t)idx->dpad * 4 > qp_row_bytes ? (int64_t)idx->dpad * 4 : qp_row_bytes;
    if (row_bytes < idx->ws_qp_row_bytes) row_bytes = idx->ws_qp_row_bytes;
    if (nq_pad > idx->ws_q || row_bytes > idx->ws_qp_row_bytes) {
        const int64_t rows = nq_pad > idx->ws_q ? nq_pad : idx->ws_q;
        LEMON_HIP_CHECK(hipStreamSynchronize(stream));
        if (idx->ws_qp) (void)hipFree(idx->ws_qp);
        if (idx->ws_qnorm) (void)hipFree(idx->ws_qnorm);
        idx->ws_qp = nullptr; idx->ws_qnorm = nullptr; idx->ws_q = 0; idx->ws_qp_row_bytes = 0;
        if (hipMalloc(&idx->ws_qp, (size_t)rows * row_bytes) != hipSuccess ||
            hipMalloc(&idx->ws_qnorm, (size_t)rows * 3 * sizeof(float)) != hipSuccess) {   // norms + bf16 residual stats
            lemon_set_error("search workspace allocation failed (nq_pad=%lld)", (long long)nq_pad);
            return LEMON_E_NOMEM;
        }
        idx->ws_q = rows; idx->ws_qp_row_bytes = row_bytes;
    }
    if (cand_rows > idx->ws_cand_rows) {
        LEMON_HIP_CHECK(hipStreamSynchronize(stream));
        if (idx->ws_cand) (void)hipFree(idx->ws_cand);
        idx->ws_cand = nullptr; idx->ws_cand_rows = 0;
        if (hipMalloc(&idx->ws_cand, (size_t)cand_rows * CAP * sizeof(u64)) != hipSuccess) {
            lemon_set_error("candidate workspace allocation failed (%lld rows)", (long long)cand_rows);
            return LEMON_E_NOMEM;
        }
        idx->ws_cand_rows = cand_rows;
    }
    if (part_elems > idx->ws_part_elems) {
        LEMON_HIP_CHECK(hipStreamSynchronize(stream));
        if (idx->ws_part) (void)hipFree(idx->ws_part);
        idx->ws_part = nullptr; idx->ws_part_elems = 0;
        if (hipMalloc(&idx->ws_part, (size_t)part_elems * sizeof(u64)) != hipSuccess) {
            lemon_set_error("merge workspace allocation failed");
            return LEMON_E_NOMEM;
        }
        idx->ws_part_elems = part_elems;
    }
    return LEMON_OK;
}

// queries are processed in chunks so that the workspace stays bounded (1 GiB of candidates)
static const int64_t QCHUNK = 1 << 19;

namespace {
// k > LEMON_MAX_K: one pass's [nq, kp] block goes to columns [col0, col0+kp) of the [nq, k] result, and the key of its
// last entry becomes the query's exclusive upper bound for the next pass (0 = the database is exhausted)
__global__ void k_place_pass(const float *__restrict__ Dp, const int64_t *__restrict__ Ip, int64_t nq, int kp, int k, int col0,
                             int metric, float *__restrict__ D, int64_t *__restrict__ I, u64 *__restrict__ ub) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * kp) return;
    const int64_t q = t / kp;
    const int c = (int)(t - q * kp);
    const float dv = Dp[t];
    const int64_t iv = Ip[t];
    D[q * k + col0 + c] = dv;
    I[q * k + col0 + c] = iv;
    if (c == kp - 1) ub[q] = iv < 0 ? 0ull : lemon_make_key(metric == LEMON_METRIC_L2 ? -dv : dv, (u32)iv);
}
__global__ void k_fill_u64(u64 *p, int64_t n, u64 v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
}  // namespace

static int lemon_search_f32_pass(lemon_index_t *idx, const float *q_dev, int64_t nq, int k, float *D_dev, int64_t *I_dev,
                                 const u64 *ub_dev, hipStream_t stream);

int lemon_search_f32(lemon_index_t *idx, const float *q_dev, int64_t nq, int k, float *D_dev,
                     int64_t *I_dev, hipStream_t stream) {
    if (k <= LEMON_MAX_K) return lemon_search_f32_pass(idx, q_dev, nq, k, D_dev, I_dev, nullptr, stream);
    // faiss has no k limit: deeper lists are produced LEMON_MAX_K at a time, each pass admitting only rows whose key is
    // below the last key of the previous pass (same order, same tie rule: the concatenation IS the sorted top-k)
    const int kp_max = LEMON_MAX_K;
    float *Dp = nullptr; int64_t *Ip = nullptr; u64 *ub = nullptr;
    const int64_t nq_pad = round_up(nq, BQ);
    if (hipMalloc((void **)&Dp, (size_t)nq * kp_max * 4) != hipSuccess || hipMalloc((void **)&Ip, (size_t)nq * kp_max * 8) != hipSuccess ||
        hipMalloc((void **)&ub, (size_t)nq_pad * 8) != hipSuccess) {
        if (Dp) (void)hipFree(Dp);
        if (Ip) (void)hipFree(Ip);
        lemon_set_error("deep search (k=%d) workspace allocation failed", k);
        return LEMON_E_NOMEM;
    }
    hipLaunchKernelGGL(k_fill_u64, dim3((unsigned)((nq_pad + 255) / 256)), dim3(256), 0, stream, ub, nq_pad, ~0ull);
    int rc = LEMON_OK;
    for (int col0 = 0; col0 < k && rc == LEMON_OK; col0 += kp_max) {
        const int kp = k - col0 < kp_max ? k - col0 : kp_max;
        rc = lemon_search_f32_pass(idx, q_dev, nq, kp, Dp, Ip, col0 ? ub : nullptr, stream);
        if (rc) break;
        hipLaunchKernelGGL(k_place_pass, dim3((unsigned)((nq * kp + 255) / 256)), dim3(256), 0, stream, Dp, Ip, nq, kp, k, col0,
                           idx->metric, D_dev, I_dev, ub);
    }
    (void)hipStreamSynchronize(stream);
    (void)hipFree(Dp); (void)hipFree(Ip); (void)hipFree(ub);
    idx->last.k = k;
    return rc;
}

static int lemon_search_f32_pass(lemon_index_t *idx, const float *q_dev, int64_t nq, int k, float *D_dev, int64_t *I_dev,
                                 const u64 *ub_dev, hipStream_t stream) {
    const int d = idx->d, dpad = idx->dpad;
    if (idx->n == 0) return lemon_fill_empty(D_dev, I_dev, nq * k, idx->metric, stream);
    const int n_tiles = (int)((idx->n + BX - 1) / BX);
    for (int64_t c0 = 0; c0 < nq; c0 += QCHUNK) {
        const int64_t cn = (nq - c0) < QCHUNK ? (nq - c0) : QCHUNK;
        const int64_t nq_pad = round_up(cn, BQ);
        const int panels = (int)(nq_pad / BQ);
        int splits, tiles_per_split = n_tiles, pieces_off = 0, segs_off = 0;
        unsigned grid;
        static const bool legacy = getenv("LEMON_SPLITS") != nullptr;   // tuning knob: (panel, split) rectangles, no plan
        const bool planned = !legacy && (int64_t)panels * n_tiles >= 2;
        const lemon_index::PlanSlot *plan = nullptr;
        if (planned) {
            const int prc = lemon_get_plan(idx, panels, n_tiles, stream, &plan);
            if (prc) return prc;
            grid = (unsigned)plan->grid; splits = plan->splits; pieces_off = plan->pieces_off; segs_off = plan->segs_off;
        } else {
            lemon_plan_splits(panels, n_tiles, &splits, &tiles_per_split);
            grid = (unsigned)(panels * splits);
        }

        int rc = lemon_ensure_search_ws(idx, nq_pad, splits, grid, dpad * 4, PAIR_CAP, stream);
        if (rc) return rc;
        // permuted, zero-padded query panel (+ chain norms for L2)
        LEMON_HIP_CHECK(hipMemsetAsync(idx->ws_qp, 0, (size_t)nq_pad * dpad * sizeof(float), stream));
        rc = lemon_permute_rows(q_dev + c0 * d, cn, d, idx->ws_qp, dpad, stream);
        if (rc) return rc;
        if (idx->metric == LEMON_METRIC_L2) {
            LEMON_HIP_CHECK(hipMemsetAsync(idx->ws_qnorm, 0, (size_t)nq_pad * sizeof(float), stream));
            rc = lemon_rowdot_chain(q_dev + c0 * d, q_dev + c0 * d, cn, d, idx->ws_qnorm, stream);
            if (rc) return rc;
        }
        ScanParams p;
        p.qp = idx->ws_qp; p.xp = idx->xp; p.qnorm = idx->ws_qnorm; p.xnorm = idx->xnorm;
        p.cand = idx->ws_cand; p.part = idx->ws_part;
        p.D = D_dev + c0 * k; p.I = I_dev + c0 * k;
        p.nq = cn; p.n = idx->n; p.dpad = dpad; p.kk = k; p.metric = idx->metric;
        p.n_tiles = n_tiles; p.tiles_per_split = tiles_per_split; p.splits = splits; p.nq_pad = nq_pad;
        p.plan = planned ? plan->dev : nullptr; p.plan_segs = segs_off; p.phase_dbg = nullptr;
        p.ub = ub_dev ? ub_dev + c0 : nullptr;
        // shared admission bounds: one order-encoded float per query, in the tail of the norm workspace ([ws_q, 3] floats:
        // norms, then two spare columns), zeroed per launch; only when a panel really is scanned by several workgroups
        static const bool share = [] { const char *e = getenv("LEMON_SHARE_BOUNDS"); return !(e && e[0] == '0'); }();
        p.th_pub = nullptr;
        if (share && splits > 1) {
            p.th_pub = reinterpret_cast<unsigned *>(idx->ws_qnorm + 2 * idx->ws_q);
            LEMON_HIP_CHECK(hipMemsetAsync(p.th_pub, 0, (size_t)nq_pad * sizeof(unsigned), stream));
        }
        // diagnostic instantiation only (LEMON_PHASE_PROF): 1 no operand loads, 2 no barriers, 4 filter off after 5 tiles,
        // 8 appends forgotten, 16 bare read-out, 32 / 128 instruction-class probes.  None of them skips maintenance.
        rc = lemon_parse_ablate("k_scan_f32", 1 | 2 | 4 | 8 | 16 | 32 | 128, &p.ablate);
        if (rc) return rc;
        static const int stale = [] { const char *e = getenv("LEMON_STALE"); return e && atoi(e) > 0 ? atoi(e) : 96; }();
        p.stale = stale;
        // fair-share turns (see k_scan_f32): about eight turns per launch, between 82 us and 5.2 ms each -- longer turns
        // measured better (16.9 / 16.7 / 16.4 / 16.3 ms at the headline shape for no turns / 5 us / 82 us / 2.6 ms), a
        // turn longer than the launch is no turn at all.  LEMON_FAIR = log2(turn in 10 ns ticks) overrides, 0 = off.
        static const int fair_env = [] { const char *e = getenv("LEMON_FAIR"); return e ? atoi(e) : -1; }();
        if (fair_env >= 0) p.fair = fair_env;
        else {
            const double ticks = 2.0 * (double)cn * (double)idx->n * (double)d / 1.25e14 * 1e8 / 8.0;   // launch / 8, in ticks
            int bit = 13;
            while (bit < 19 && (double)(2u << bit) <= ticks) ++bit;
            p.fair = bit;
        }
        {
            const double flops = 2.0 * (double)cn * (double)idx->n * (double)d;
            const double bytes = 4.0 * d * ((double)cn + (double)panels * (double)idx->n) + 12.0 * k * (double)cn;
            LemonProfScope prof(idx, stream, flops, bytes);
            if (!ub_dev && idx->metric == LEMON_METRIC_IP && getenv("LEMON_PHASE_PROF")) {   // diagnostic build: phase cycle sums
                static unsigned long long *dbg = nullptr;
                if (!dbg) { (void)hipMalloc(&dbg, 128 + 8 * 4096); (void)hipMemset(dbg, 0, 128 + 8 * 4096); }
                (void)hipMemset(dbg + 7, 0xff, 8);
                p.phase_dbg = dbg;
                hipLaunchKernelGGL((k_scan_f32<false, true>), dim3(grid), dim3(NT), 0, stream, p);
                (void)hipStreamSynchronize(stream);
                unsigned long long h[16]; (void)hipMemcpy(h, dbg, 128, hipMemcpyDeviceToHost);
                const double tot = (double)(h[0] + h[1] + h[2] + h[3]);
                fprintf(stderr, "[phase f32] grid=%u loop=%.1f%% filter=%.1f%% maintain=%.1f%% final=%.1f%% cyc/WG=%.3g\n",
                        grid, 100.0 * h[0] / tot, 100.0 * h[1] / tot, 100.0 * h[2] / tot, 100.0 * h[3] / tot, tot / grid);
                fprintf(stderr, "[clock f32] workgroup 0: %llu shader cycles in %llu ticks of the 100 MHz counter = %.0f MHz\n",
                        h[4], h[5], h[5] ? 100.0 * (double)h[4] / (double)h[5] : 0.0);
                fprintf(stderr, "[lives f32] workgroup life min %.2f ms  max %.2f ms;  wave-slot parity 0: %llu workgroups, mean %.2f ms;  parity 1: %llu, mean %.2f ms\n",
                        h[7] / 1e5, h[6] / 1e5, h[9], h[9] ? h[8] / 1e5 / h[9] : 0.0, h[11], h[11] ? h[10] / 1e5 / h[11] : 0.0);
                if (const char *dump = getenv("LEMON_LIVES_DUMP")) {      // per-workgroup lives (100 MHz ticks), one per line
                    std::vector<unsigned long long> lv(grid < 4096 ? grid : 4096);
                    (void)hipMemcpy(lv.data(), dbg + 16, lv.size() * 8, hipMemcpyDeviceToHost);
                    if (FILE *f = fopen(dump, "w")) { for (auto v : lv) fprintf(f, "%llu\n", v); fclose(f); }
                }
                (void)hipMemset(dbg, 0, 128 + 8 * 4096);
            } else if (ub_dev) {
                if (idx->metric == LEMON_METRIC_L2) hipLaunchKernelGGL((k_scan_f32<true, false, true>), dim3(grid), dim3(NT), 0, stream, p);
                else hipLaunchKernelGGL((k_scan_f32<false, false, true>), dim3(grid), dim3(NT), 0, stream, p);
            } else if (idx->metric == LEMON_METRIC_L2) hipLaunchKernelGGL((k_scan_f32<true, false>), dim3(grid), dim3(NT), 0, stream, p);
            else hipLaunchKernelGGL((k_scan_f32<false, false>), dim3(grid), dim3(NT), 0, stream, p);
        }
        LEMON_HIP_CHECK(hipGetLastError());
        if (splits > 1) {
            rc = lemon_launch_merge(idx->ws_part, splits, planned ? plan->dev + pieces_off : nullptr, nq_pad, cn, k, idx->metric, p.D,
                                    p.I, stream);
            if (rc) return rc;
        }
        idx->last.algo = LEMON_ALGO_F32_MFMA;
        idx->last.grid = (int)grid; idx->last.block = NT;
        idx->last.query_panel = BQ; idx->last.db_splits = splits;
    }
    idx->last.nq = nq; idx->last.n = idx->n; idx->last.d = d; idx->last.k = k;
    return LEMON_OK;
}
