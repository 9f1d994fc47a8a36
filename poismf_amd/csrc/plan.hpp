// plan.hpp -- what the host side (poismf_hip_host.hip) and the row-kernel translation units (poismf_hip.hip, compiled once
// per inner solver) share: the kernel argument block, the description of one planned launch, and the constants / small
// functions that decide which engine and which instance a row-length bin takes.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "../../include/poismf_hip.h"
#include "row_eval.hpp"
#include "devmem.hpp"

using namespace pmf;

// The reference's ABI has one failure code (1, "out of memory", ref: src/poismf.c:498-504).  The callers of this library get
// that code for every failure too, but stderr says what happened: the last HIP error of the calling thread is kept here
// and the entry points print the reference's message only when it was an allocation that failed.
inline hipError_t& pmf_last_hip_error()
{
    static thread_local hipError_t e = hipSuccess;
    return e;
}
inline void pmf_report_failure()
{
    const hipError_t e = pmf_last_hip_error();
    if (e == hipSuccess || e == hipErrorOutOfMemory) fprintf(stderr, "Error: out of memory.\n");   // ref: src/poismf.c:501
    else fprintf(stderr, "Error: the HIP device or runtime failed (%s) -- not an out-of-memory condition; the factors are not valid.\n", hipGetErrorString(e));
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            pmf_last_hip_error() = e_;                                                             \
            fprintf(stderr, "poismf_hip: %s failed: %s\n", #expr, hipGetErrorString(e_));          \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

struct RowDesc;
template <class T> struct HalfArgs {
    T* M;                             // factor being updated, [dimM x k]
    T* Mp;                            // its line-padded copy (row stride ldM), or nullptr: updated rows go to both
    int ldM;
    const T* F;                       // opposing factor, [dimF x k] (+16 B of slack)
    const unsigned long long* indptr; // shard-local CSR row pointers (nrows_local + 1)
    const unsigned* indices;
    const T* values;
    const unsigned* perm;             // shard-local row ids, sorted by nnz descending
    const struct RowDesc* desc;       // the same order, with each row's CSR offset and length (one load per row)
    unsigned perm_begin, nrows;       // this launch covers perm[perm_begin, perm_begin + nrows)
    unsigned row_offset;              // first global row of the shard (M row = row_offset + local id)
    const T* bsum;                    // k-vector: colsum(F) + l1 (pre-scaled for PG when w == 1)
    TileGeom geom;
    RowParams<T> P;
    int reuse_prev, early_stop;
    unsigned* n_unchanged;
    unsigned* queue;                  // != nullptr: rows are handed out dynamically through this counter
    const unsigned* stop;             // != nullptr (CG / TNCG): a word in pinned host memory that the SIGINT handler sets -- read next to every row ticket, as
                                      // the reference's row loops read should_stop_procedure before every row (ref: src/poismf.c:301, :360)
    unsigned* eval_rows;              // != nullptr (profiling sessions): [local row] += passes over that row's tile
    unsigned* dec_rows;               // != nullptr (profiling sessions): [2 x local row] = the solver's decisions (solvers.hpp, SolveStats)
    unsigned long long* team_buf;     // team launches (several CUs per row, reg_eval.hpp M_ > 1): arrival counters, mailboxes, exchange slots
    unsigned* team_err;               // set by a team launch that gave up (an exchange timed out); the host re-runs such a launch
    unsigned team_spin;               // polls (~1 us each) before a team member gives the launch up (TEAM_SPIN_LIMIT; a knob for tests)
    unsigned team_members;            // giant-row / lane teams (row_eval.hpp TM, lane_eval.hpp TM_): workgroups per row
    const unsigned* gate;             // != nullptr: the kernel runs only if *gate != 0 (the streamed re-run of a team launch that gave up)
    unsigned* arrive;                 // != nullptr (the long-row launch on the second stream): every workgroup counts itself in here when it
                                      // starts -- the main stream holds the other bins' kernels back until the long rows are on the chip
};

enum { K_PG = 3, K_CG = 2, K_TNCG = 1, K_EVAL = 4 };
// K_EVAL / POISMF_EVAL: not a solver -- the row kernels evaluate the device's own fun_single + grad_single (RowParams::maxupd == 0) or
// fun_and_grad (maxupd == 1) at each row's starting point and hand the gradient back in place of the updated row, the function
// value through the decisions words (testing aid: poismf_hip_debug_row_eval, include/poismf_hip.h).  Planned like CG, so every
// engine CG would pick for a row length is reachable; its kernels live in a translation unit of their own (PMF_TU=4).
constexpr int POISMF_EVAL = 4;

// One row of the sorted order: where its nonzeros start in the shard's CSR arrays, how many, and which row it is.
struct RowDesc { unsigned p0_lo, p0_hi, nnz, lrow; };

namespace {

constexpr size_t LDS_PER_CU = 160 * 1024;

// Slot counts with a compile-time specialisation: the k values of the BASELINE configs
// (fp32: k = 49..52 -> 13 slots, k = 97..100 -> 25; fp64: k = 49..50 -> 25, k = 99..100 -> 50).
#ifdef USE_FLOAT
constexpr int SPECIAL_SL_A = 13, SPECIAL_SL_B = 25;
#else
constexpr int SPECIAL_SL_A = 25, SPECIAL_SL_B = 50;
#endif

// ---- register-tile engine (reg_eval.hpp) ---------------------------------------------------------------------------
// A factor row is held by REG_G lanes (NS = 1 or 2 slots per lane); a step covers 64 / REG_G nonzeros.  Tile steps S
// with an instantiated kernel; a bin takes the smallest S that covers its longest row.
#ifndef PMF_REG_G
#define PMF_REG_G 16
#endif
constexpr int REG_G = PMF_REG_G;
constexpr int REG_JG = WAVE / REG_G;
// tile sizes in nonzeros with an instantiated kernel (S = nonzeros / REG_JG steps)
#define PMF_REG_SIZES(X) X(16) X(32) X(48) X(64) X(80) X(96) X(112) X(128) X(144) X(160)
constexpr int REG_NNZ_MAX = 160;
// 16-byte slots per lane: factor rows of up to REG_G slots take one, doubles with up to 2 REG_G slots (k = 50 fp64: 25) take
// two -- one wave per row only, at one wave per SIMD (the tile alone is 8 S registers); 0 = not a register-engine row.
constexpr int REG_NS_MAX = (sizeof(real_t) == 8 && REG_G == 16) ? 2 : (REG_G == 8 ? 2 : 1);
inline int reg_slots_per_lane(size_t s_load) { return s_load <= (size_t)REG_G ? 1 : (s_load <= (size_t)REG_G * REG_NS_MAX ? 2 : 0); }
int reg_steps_for(unsigned max_nnz)
{
#define X(NZ) if ((unsigned)(NZ) >= max_nnz) return (NZ) / REG_JG;
    PMF_REG_SIZES(X)
#undef X
    return 0;
}

// longest row (nonzeros) each solver runs from a register tile; beyond it the LDS engine has more waves per CU
#ifndef PMF_REG_MAX_CG
#define PMF_REG_MAX_CG 160
#endif
#ifndef PMF_REG_MAX_TNCG
#define PMF_REG_MAX_TNCG 160
#endif
constexpr int REG_NNZ_MAX_CG = PMF_REG_MAX_CG, REG_NNZ_MAX_TNCG = PMF_REG_MAX_TNCG;
unsigned reg_nnz_max(int method) { return method == POISMF_PG ? REG_NNZ_MAX : method == POISMF_CG ? REG_NNZ_MAX_CG : REG_NNZ_MAX_TNCG; }

constexpr int REG_NW_MAX = 8;   // the same engine with 2, 4 or 8 wavefronts per row
// Several waves per row: the longest share of a row one wave keeps in registers, per solver (CG / TNCG carry more state)
// (CG: 144 although the 32- and 36-step fp32 instances spill 200-300 bytes per lane at two waves per SIMD -- for rows of
// 1025-1152 nonzeros the alternative is the streamed LDS path: C4's matrix, CG fp32, B half 25.0 -> 16.8 ms)
constexpr int REGW_WAVE_NNZ_MAX_PG = 160, REGW_WAVE_NNZ_MAX_CG = 144, REGW_WAVE_NNZ_MAX_TNCG = 96;
unsigned regw_wave_nnz_max(int method)
{
    return (unsigned)(method == POISMF_PG ? REGW_WAVE_NNZ_MAX_PG : method == POISMF_CG ? REGW_WAVE_NNZ_MAX_CG : REGW_WAVE_NNZ_MAX_TNCG);
}
unsigned regw_nnz_max(int method) { return (unsigned)REG_NW_MAX * regw_wave_nnz_max(method); }
// Sixteen waves per row (one 1024-thread workgroup, four waves per SIMD, <= 128 registers each: shares of <= 64
// nonzeros) for rows of 513 .. 1024 nonzeros: measured and NOT adopted (-DPMF_REGW16=1 builds it).  One such row occupies
// a CU either way and a SIMD gets four instruction streams instead of two, but the per-pass fixed work of a wave (point
// update, group combine, reading 16 partial gradients, a 16-wave barrier) is paid twice as often: C4 matrix, PG(10), B half
// 7.9 ms with eight waves per row, 13.9 ms with sixteen.
#ifndef PMF_REGW16
#define PMF_REGW16 0
#endif
constexpr int REGW16_WAVE_NNZ = 64;
inline bool regw16_method(int method) { return PMF_REGW16 && method == POISMF_PG; }
// the fewest waves (2, 4, 8) whose shares of a row of max_nnz nonzeros fit; 16 where that pays (see above)
int regw_waves_for(unsigned max_nnz, int method)
{
    if (regw16_method(method) && max_nnz > 8u * REGW16_WAVE_NNZ && max_nnz <= 16u * REGW16_WAVE_NNZ) return 16;
    for (int nw : { 2, 4, 8 })
        if (max_nnz <= (unsigned)nw * regw_wave_nnz_max(method)) return nw;
    return 0;
}
// tile steps for a row of max_nnz nonzeros split over nw waves (each wave's share is rounded up to whole steps)
int regw_steps_for(unsigned max_nnz, int nw)
{
    const unsigned share = ((max_nnz + (unsigned)nw - 1) / (unsigned)nw + REG_JG - 1) / REG_JG * REG_JG;
    return reg_steps_for(std::max(32u, share));
}

// teams (row_eval.hpp: TEAM_*): members and tile steps for rows of up to max_nnz nonzeros (members 0: not a team row).
// Two CUs where they hold the row; the 28-step instance (no scratch) where three are needed anyway.
// Below TEAM_MIN_NNZ a row's tile (k = 50 fp64: 400 B per nonzero) is resident in one CU's LDS: the LDS engine keeps it.
constexpr unsigned TEAM_MIN_NNZ = 385;
struct TeamShape { int members, steps; };
inline TeamShape team_shape_for(unsigned max_nnz)
{
    const unsigned per_step = (unsigned)(REG_JG * TEAM_NW);
    if (max_nnz < TEAM_MIN_NNZ) return { 0, 0 };
    if (max_nnz <= 2u * 32u * per_step) return { 2, 32 };
    if (max_nnz <= 2u * 36u * per_step) return { 2, 36 };
    if (max_nnz <= 3u * 28u * per_step) return { 3, 28 };
    if (max_nnz <= 3u * 32u * per_step) return { 3, 32 };
    if (max_nnz <= 4u * 32u * per_step) return { 4, 32 };
    return { 0, 0 };
}

// Lane-per-nonzero engine (lane_eval.hpp): factor rows of 25 slots in doubles (k = 49..50; 50 slots, k = 99..100, for the
// shortest rows) and of 13 slots in floats (k = 49..52).  Lane sets (64 nonzeros each) per wave -- in architectural registers,
// in accumulator registers, in LDS -- and waves per row for rows of a length class; waves 0 = not a row of this engine.
// A function of the class bound (and the solver) alone, so a row's arithmetic does not depend on its shard.
// lv / la / ll: lane sets per wave in architectural registers / accumulator registers / LDS; waves: per row; small: the two-waves-per-SIMD flavour (a few KB
// of LDS per wave); lp: nonzeros of a further, partial LDS set; tx: rows of the LDS image the gradient is accumulated from (lane_eval.hpp, TX_)
struct LaneShape { int lv, la, ll, waves; int small; int lp = 0; int tx = 0; };
inline LaneShape lane_shape_for(unsigned cls, int s_load, int method)
{
    if (sizeof(real_t) == 8) {
        if (method == POISMF_PG) return { 0, 0, 0, 0, 0 };
        if (s_load == 25) {          // a set is 100 registers / 25.6 KB of LDS
            if (cls <= 64) return { 1, 0, 0, 1, 1 };
            // 65 .. 96 nonzeros (round 5; a third of config C3's user rows): one register set + a PARTIAL LDS set of 32 -- 18.8 KB of LDS per row
            // instead of the 36 KB of a full LDS set: eight rows per CU, two waves per SIMD
            if (cls <= 96 && method != POISMF_TNCG) return { 1, 0, 0, 1, 1, 32 };
            // (97 .. 112 nonzeros on a partial set of 48 -- six rows per CU instead of the full LDS set's four -- and 65 .. 128 on two waves of one register
            // set each were measured in round 6: 12.59 / 12.66 ms against 12.63 for the A half's 626 801 such rows of C3, DESIGN.md 6.0c: not kept)
            if (cls <= 128) return LaneShape{ 1, 0, 1, 1, 0 };
            if (cls <= 256) return { 1, 2, 1, 1, 0 };
            if (cls <= 512) return { 1, 2, 1, 2, 0 };
            if (cls <= 1024) return { 1, 2, 1, 4, 0 };
            if (cls <= 1088) return { 1, 2, 1, 4, 0, 16 };   // 4 x (64 + 128 + 64 + 16) nonzeros: C3's item rows (Poisson(1000)) end here but for 0.3 %
        } else if (s_load == 50) {   // a set is 200 registers / 51 KB of LDS
            // rows of at most 64 nonzeros: the gradient from a second, row-major copy of the tile in LDS instead of the transposing
            // reduction (lane_eval.hpp, TX_): 39 KB of LDS for up to 48 nonzeros (four rows per CU), 52 KB up to 64 (three).
            // (the transposing reduction for these rows -- rounds 3-4, POISMF_HIP_NO_TX -- and the image for 48 nonzeros only went in round 6)
            if (cls <= 48) return { 1, 0, 0, 1, 0, 0, 48 };
            if (cls <= 64) return { 1, 0, 0, 1, 0, 0, 64 };
            // 65 .. 128 nonzeros: one register set + one LDS set on one wave takes 63 KB of LDS -- two rows per CU, two of its four SIMDs idle
            // (rounds 4-5a); two waves of one register set each (no LDS set) keep all four busy on the same two rows
            if (cls <= 128) return LaneShape{ 1, 0, 0, 2, 0 };
            // 129 .. 384 nonzeros (round 5): four waves of one register set + a partial LDS set of 32 nonzeros each, one row per CU: config
            // C5's item rows of this length stay on chip for all of TNC's ~70 evaluations instead of re-streaming 800 bytes per nonzero
            // for each of them (153 x the algorithmic traffic in round 4's streamed launch)
            if (cls <= 384) return { 1, 0, 0, 4, 0, 32 };
        }
    } else if (s_load == 13) {   // floats: a set is 52 registers, every set in architectural registers, two waves per SIMD
        if (method == POISMF_PG) {
            // PG does the same work on every pass and the slot layout (reg_eval.hpp) is the cheaper one for single-wave rows; what
            // the lane layout buys is the long rows: eight waves with NO cross-lane traffic in the dots and one transposing
            // reduction per wave instead of a butterfly per four-nonzero step
            // (a first port -- two or three sets per wave, eight waves per row, PMF_LANE_PG32 / POISMF_HIP_PG_LONG_LANE -- lost to reg_eval.hpp's eight-wave
            // kernels, 9.5 against 8.0 ms for the B half and 1.93 against 1.83 ms on the rows above 1024; both went in round 6)
            // rows of 513 .. 1024 nonzeros on FOUR waves of four register sets each, two such rows per CU
            if (cls > 512 && cls <= 1024) return { 4, 0, 0, 4, 1 };
            // 1025 .. 1088 nonzeros (round 6; 98.6 % of the C4 matrix's item rows above 1024: Poisson(1000)): the same four waves of four register sets
            // + a PARTIAL LDS set of 16 nonzeros per wave (4 KB; 76 KB of LDS per row: still two rows per CU) instead of the register engine's
            // eight-wave kernel at one row per CU (0.32 of the byte roofline, the worst bucket of round 5's headline: 1.83 -> 1.21 ms, 0.47)
            if (cls > 1024 && cls <= 1088) return { 4, 0, 0, 4, 1, 16 };
            // (the A half's one-wave rows on this engine's one- and two-set instances, measured in round 6: the headline sweep 9.29 -> 10.75 ms -- the
            // slot layout of reg_eval.hpp stays the cheaper one for single-wave PG rows; not kept)
            return { 0, 0, 0, 0, 0 };
        }
        if (cls <= 64) return { 1, 0, 0, 1, 1 };
        if (cls <= 128) return { 2, 0, 0, 1, 1 };
        if (cls <= 256) return { 2, 0, 0, 2, 1 };
        if (cls <= 512) return { 2, 0, 0, 4, 1 };
        if (cls <= 1024) return { 2, 0, 0, 8, 1 };
        if (cls <= 1536) return { 3, 0, 0, 8, 1 };
    }
    return { 0, 0, 0, 0, 0 };
}

// Long-row path: rows above this many nonzeros get a whole workgroup of LONG_NW waves (row_eval.hpp, NW > 1).
constexpr unsigned LONG_ROW_NNZ = 8192;
constexpr int LONG_NW = 8;
constexpr int SLOT_ELEMS = (int)(16 / sizeof(real_t));   // elements per 16-byte slot

// row kernels: 16-byte slots per lane (slot layout of row_eval.hpp); 0 = unsupported
int slots_per_lane(size_t k)
{
    const size_t s_load = (k * sizeof(real_t) + 15) / 16;
    return s_load <= 64 ? 1 : (s_load <= 128 ? 2 : 0);
}

}  // namespace

// One row-bin launch: everything the planner decided, minus the solver (which selects the translation unit).
struct OneLaunch {
    int reg_S, nw, s_load, spl;   // register-engine steps (0: LDS engine), waves per row, slots per factor row, slots per lane
    int team;                     // > 1: CUs per row (team launch)
    int lane_LP;                  // lane engine: nonzeros of the partial LDS set per wave
    int lane_small;               // lane engine: the two-waves-per-SIMD flavour (lane_eval.hpp, SMALL_)
    int lane_tx;                  // lane engine: rows of the LDS image of the tile (lane_eval.hpp, TX_)
    int lane_L, lane_A, lane_LL;  // lane_L > 0: lane-per-nonzero engine with this many lane sets per wave in VGPRs, AGPRs, LDS (nw waves per row)
    bool generic_only;
    hipStream_t main_stream, bin_stream, long_stream;
    size_t lds;
    unsigned grid, grid_mult;
    int device, num_cu;
};


// One function per row-kernel translation unit (PMF_TU = 1 tncg, 2 cg, 3 pg): launches the instance the plan names.
int pmf_launch_one_tu1(int method, const OneLaunch& o, const HalfArgs<real_t>& a);
int pmf_launch_one_tu2(int method, const OneLaunch& o, const HalfArgs<real_t>& a);
int pmf_launch_one_tu3(int method, const OneLaunch& o, const HalfArgs<real_t>& a);
int pmf_launch_one_tu4(int method, const OneLaunch& o, const HalfArgs<real_t>& a);
