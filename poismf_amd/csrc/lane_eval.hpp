// lane_eval.hpp -- the per-row evaluation engine for DOUBLES: one nonzero per LANE, its whole factor row in that lane's registers.
//
// Same contract as RegEval / RowEval: the factor rows F[ind_j] named by a row's nonzeros are fetched once and every inner
// pass of the solver runs on chip.  The layout is the transpose of reg_eval.hpp's:
//
//   tile     t[s][c], s < LT "lane sets", c < KP = 2 KS: lane l of set s holds ALL KP elements of F[ind_j], j = 64 s + l
//            (k = 50: 100 registers per set).  Nothing is padded to a power of two: 25 slots cost 25 slots (the slot
//            layout of reg_eval.hpp pays for 32).  Where a set lives is stated, not left to the register allocator (which,
//            given two sets and the solver, parks half the tile in AGPRs and copies every element out for every use --
//            measured: 43 % of the instructions of a pass): LV sets in architectural registers, LA sets in accumulator
//            registers through explicit v_accvgpr_write / read (two reads per element and pass: for rows that outgrow
//            everything else), LL sets in LDS -- the staged image of the gather simply stays where the DMA put it and is read
//            with one conflict-free ds_read_b128 per slot (half an instruction per element and pass).
//   k-vector lane <-> dimension, ONE copy per wave (NC = 1 element per lane for k <= 52, 2 for k <= 104): the solver's whole
//            state is 2 NC registers per vector instead of 8, every element-wise statement of the solvers is one
//            instruction, and nothing is computed four times over (the slot layout keeps four copies per wave).
//            Which lane holds which dimension is chosen for the reduction below: with DB = KP / NC dimensions per block and
//            CW = ceil(DB / 4) columns, dimension d' of a block sits in lane (d' % CW) + 16 (d' / CW) -- k = 50: lanes 0-12
//            hold dimensions 0-12, lanes 16-28 13-25, lanes 32-44 26-38, lanes 48-58 39-49.
//   dots     pred_j = F[ind_j] . a: a plain chain of KP fused multiply-adds per lane, in the reference's own summation
//            order (ddot, left to right); `a` is read as a broadcast from LDS, one ds_read_b128 per two elements for all
//            lane sets.  No cross-lane traffic at all: the dot of nonzero j is finished in the lane that owns j.
//   axpy     g_c = sum_j coef_j F[ind_j, c]: every lane forms coef_j t[s][c] for its own nonzeros (KP multiply-adds per
//            set), and the KP lane-partials are summed over the 64 lanes by a TRANSPOSING reduction that leaves the total
//            of a dimension in the lane that holds it: per column, the four dimensions col + CW r meet in two levels of
//            v_permlane32_swap / v_permlane16_swap folds (three instructions per fold on doubles; the lane map above makes
//            every fold a full pair: 37 folds for 50 dimensions), then a CW x 16 transpose-and-add per 16-lane row through
//            LDS (CW ds_write_b64, 8 conflict-free ds_read_b128, 15 adds).
//   gather   the tile is requested with LDS-DMA (global_load_lds_dwordx4: 64 consecutive 16-byte slots of the row-major
//            image [nonzero][13 slots] per instruction, every lane's source address its own), so the fetch is as
//            coalesced as reg_eval.hpp's (13 slots = 208 contiguous bytes per nonzero) and costs no registers; each lane
//            then pulls its own nonzero's slots out of the image with ds_read_b128 (row stride 208 B: conflict-free).
//
// Instruction count of one gradient pass over a 128-nonzero tile, k = 50: 200 fp64 FMAs (dots + axpy), 144 swap-fold
// instructions, 15 adds, ~60 for the two divisions, ~50 LDS operations -- ~470 against ~950 in the slot layout
// (34 instructions per four-nonzero step, 16 of them v_accvgpr_read).
//
// This is the reference's per-nonzero ddot + daxpy (ref: src/poismf.c:126-133 calc_grad_pgd, :194-208
// calc_fun_single, :210-240 calc_grad_single[_w], :242-273 calc_fun_and_grad).
#pragma once
#include <type_traits>

#include "reg_eval.hpp"

namespace pmf {

// slots per group of the dots' software pipeline (eval(): the point and the LDS sets' slots of the NEXT group are requested before
// the current group's multiply-adds: 2 x GQ x (1 + LDS sets) 16-byte values in flight).  Five where registers allow it; the instances
// with two AGPR sets are at 512 registers, and the buffers of five-slot groups were what sent them to scratch (round 3: 68 / 96 / 96 /
// 420 bytes per lane for one / two / four waves per row / with the partial set) -- three slots, two with the partial set: 0 / 0 / 0 / 56.
#ifndef PMF_LANE_GQ
#define PMF_LANE_GQ(la, lp) ((la) >= 2 ? ((lp) > 0 ? 2 : 3) : 5)
#endif

// a + b where lanes with bit 5 (W = 32) / bit 4 (W = 16) clear collect `a` and the others collect `b`: every lane ends
// with its own copy of the operand it collects plus the partner lane's (lane ^ W) copy of the same operand.
template <int W> __device__ __forceinline__ double swap_fold(double a, double b)
{
    const unsigned long long ab = __builtin_bit_cast(unsigned long long, a), bb = __builtin_bit_cast(unsigned long long, b);
    unsigned lo0, lo1, hi0, hi1;
    if constexpr (W == 32) {
        const auto rl = __builtin_amdgcn_permlane32_swap((unsigned)ab, (unsigned)bb, false, false);
        const auto rh = __builtin_amdgcn_permlane32_swap((unsigned)(ab >> 32), (unsigned)(bb >> 32), false, false);
        lo0 = rl[0]; lo1 = rl[1]; hi0 = rh[0]; hi1 = rh[1];
    } else {
        const auto rl = __builtin_amdgcn_permlane16_swap((unsigned)ab, (unsigned)bb, false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap((unsigned)(ab >> 32), (unsigned)(bb >> 32), false, false);
        lo0 = rl[0]; lo1 = rl[1]; hi0 = rh[0]; hi1 = rh[1];
    }
    return __builtin_bit_cast(double, ((unsigned long long)hi0 << 32) | lo0) + __builtin_bit_cast(double, ((unsigned long long)hi1 << 32) | lo1);
}

template <int W> __device__ __forceinline__ float swap_fold(float a, float b)
{
    const unsigned ab = __builtin_bit_cast(unsigned, a), bb = __builtin_bit_cast(unsigned, b);
    if constexpr (W == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(ab, bb, false, false);
        return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    } else {
        const auto r = __builtin_amdgcn_permlane16_swap(ab, bb, false, false);
        return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
}

// The two upper levels of the FLOAT butterfly in place (round 6).  A fold of the transposing butterfly is "lanes of class 0 keep a and add the partner
// lane's a, lanes of class 1 keep b and add the partner's b"; through update_dpp that is two selects and one DPP add (reg_eval.hpp, fold_pair).  For the
// levels whose classes are DPP BANKS (lane ^ 8 under row_ror:8: banks {0,1} | {2,3}; lane ^ 7 under row_half_mirror: banks {0,2} | {1,3}) it is two
// v_add_f32_dpp with complementary bank masks -- as reg_eval.hpp's fold16_banked does with separate, early-clobber outputs, which the lane instances
// (253-256 VGPRs) have no registers for.  Here the sum lands IN a: the first add rewrites a's class-0 banks from a, the second its class-1 banks from
// b, and neither reads what the other wrote.  Same operands, same sums, same bits as fold_pair; one instruction per fold less (12 of a batch's 15 folds:
// 39 of the 512 instructions of the PG headline kernel's pass).  s_nop 1: a DPP source written by the previous VALU instruction needs two wait states,
// and the hazard recogniser does not look inside inline asm.
#define PMF_FOLD_IN(CTRL, M0, M1, A, B) \
    "v_add_f32_dpp " A ", " A ", " A " " CTRL " row_mask:0xf bank_mask:" M0 "\n\t" \
    "v_add_f32_dpp " A ", " B ", " B " " CTRL " row_mask:0xf bank_mask:" M1 "\n\t"
__device__ __forceinline__ void fold4_ror8_inplace(float& a0, float& a1, float& a2, float& a3, float b0, float b1, float b2, float b3)
{
    asm("s_nop 1\n\t"
        PMF_FOLD_IN("row_ror:8", "0x3", "0xc", "%0", "%4")
        PMF_FOLD_IN("row_ror:8", "0x3", "0xc", "%1", "%5")
        PMF_FOLD_IN("row_ror:8", "0x3", "0xc", "%2", "%6")
        PMF_FOLD_IN("row_ror:8", "0x3", "0xc", "%3", "%7")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}
__device__ __forceinline__ void fold4_mirror_inplace(float& a0, float& a1, float& a2, float& a3, float b0, float b1, float b2, float b3)
{
    asm("s_nop 1\n\t"
        PMF_FOLD_IN("row_half_mirror", "0x5", "0xa", "%0", "%4")
        PMF_FOLD_IN("row_half_mirror", "0x5", "0xa", "%1", "%5")
        PMF_FOLD_IN("row_half_mirror", "0x5", "0xa", "%2", "%6")
        PMF_FOLD_IN("row_half_mirror", "0x5", "0xa", "%3", "%7")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}

// Nothing moves across this line: neither loads at the IR / instruction-selection level (the memory clobber; a bare
// sched_barrier is no obstacle there, and hipcc then requests a whole pass's LDS reads at once and parks their 200 destination
// registers -- in practice: the tile -- in AGPRs) nor anything in the machine scheduler.
__device__ __forceinline__ void pin_here()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// AGPR-class values: written and read only through these (the "a" constraint makes the allocator keep them in the
// accumulator half of the register file; nothing else of the kernel lives there)
__device__ __forceinline__ unsigned acc_put(unsigned v)
{
    unsigned a;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
    return a;
}
__device__ __forceinline__ unsigned acc_get(unsigned a)
{
    unsigned v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}

// SMALL_: one staging buffer, shared with the transpose scratch (14 KB of LDS per wave: eight waves per CU, two per SIMD, for
// kernels whose registers allow that -- one set in architectural registers, nothing in AGPRs)
// LP_: a further, PARTIAL set of LP_ (16) nonzeros per wave in LDS (lanes >= LP_ of that set alias lanes < LP_ with a zero
// coefficient), and the transposing reduction's scratch used in two halves -- what makes rows of 1025 .. 1088 nonzeros fit ONE
// CU: 4 waves x (64 + 128 + 64 + 16) nonzeros, 155 KB of LDS.
// TX_ (round 5; doubles, one register set, one wave per row: k = 100 rows of at most 64 nonzeros, config C5's user rows and short item rows):
// the gradient sum_j c_j F_j is NOT formed by the transposing reduction (for 100 dimensions: 78 double swap folds at ~32 cycles each,
// 100 partial products and, because hipcc keeps the 200-register tile of these instances in AGPRs, 200 v_accvgpr_reads at 6 cycles --
// 5.2 k of an evaluation's 10 k cycles, scripts/probes/f64_probe.hip and the duplication builds of DESIGN.md 6.0b) but the way the
// reference forms it (ref: src/poismf.c:126-133, :262-267: one daxpy per nonzero, in order): the tile sits in LDS as the row-major
// image [nonzero][slots] (written once per row), every lane owns its two dimensions, and per nonzero j it takes c_j as a scalar (two
// v_readlane) and its two elements of row j (conflict-free ds_read_b64) into two multiply-adds -- six instructions per nonzero,
// nothing per dimension, the reference's own summation order.  The dots read the same image (lane j its own row, one ds_read_b128
// per slot, row stride 816 bytes = conflict-free), so NO tile lives in registers: the solver's state has the 256 architectural
// registers to itself and the v_accvgpr traffic is gone.  TX_ = rows of the image: 48 (39 KB: four rows per CU) or 64 (52 KB: three).
template <class T, int KS, int LV_, int LA_ = 0, int LL_ = 0, int NW_ = 1, bool SMALL_ = false, int LP_ = 0, int TX_ = 0, bool TM_ = false> struct LaneEval {
    using SA = typename Slot<T>::A;
    static constexpr int SN = Slot<T>::N;                 // elements per 16-byte slot
    static constexpr int KP = KS * SN;                    // elements of a factor row, padded to whole slots
    static constexpr int NC = (KP + WAVE - 1) / WAVE;     // elements of a k-vector per lane
    static constexpr int LP = LP_;                                                  // nonzeros of the partial LDS set (0: none)
    static constexpr int LLT = LL_ + (LP_ > 0 ? 1 : 0);                              // sets read from LDS (the partial one last)
    static constexpr int LV = LV_, LA = LA_, LL = LL_, LR = LV_ + LA_, LT = LV_ + LA_ + LLT;   // sets: VGPR, AGPR, LDS
    static_assert(LP_ == 0 || ((LP_ & (LP_ - 1)) == 0 && LP_ < WAVE), "a partial set: a power of two of nonzeros, the last set of the wave");
    // the row of the partial set's image that lane l reads: its own below LP, an earlier lane's (finite data, coefficient 0) from there on
    static __device__ __forceinline__ int part_row(int l) { return l & (LP_ - 1); }
    static constexpr int L = LT, NW = NW_, M = 1;
    static constexpr int W = KS < 13 ? KS : 13;           // slots per staged chunk (row stride of the LDS image: 13 slots = 52 banks, conflict-free b128 reads)
    static constexpr int NCH = (KS + W - 1) / W;          // chunks per factor row; chunk c starts at slot min(c W, KS - W)
    static constexpr int STAGE_BYTES = WAVE * W * 16;
#ifndef PMF_LANE_DIRECT
#define PMF_LANE_DIRECT 1   // the sets in architectural registers are loaded lane by lane, without staging (gather())
#endif
    static constexpr bool DIRECT_V = PMF_LANE_DIRECT;
    // (measured and gone in round 6: the AGPR sets loaded straight into their registers by inline asm -- unsafe, hipcc spills "loaded" values before
    // the data has landed; AGPR sets through free VGPRs -- C3 B half 13.2 -> 15.6 ms, a per-lane load names 64 rows = 64 tag look-ups; the VGPR set of
    // the AGPR instances staged through LDS -- within noise.  DESIGN.md 4.4 keeps the numbers.)
    static constexpr bool STAGED = LL_ > 0 || LA_ > 0 || !DIRECT_V;   // some set travels through LDS
    static constexpr int NBUF = LL_ > 0 ? LL_ * NCH : (!STAGED ? 0 : (SMALL_ ? 1 : 2));  // staging buffers; the chunks of the LDS sets stay in theirs
    static constexpr bool ALIAS = SMALL_ && NBUF > 0 && LL_ == 0;   // (with an LDS set the buffers hold the tile: the scratch gets its own bytes)
    static_assert(KP % NC == 0, "blocks of equal size");
    static constexpr int DB = KP / NC;                    // dimensions per 64-lane block
    static constexpr int CW = (DB + 3) / 4;               // columns: dimension d' of a block lives in lane (d' % CW) + 16 (d' / CW)
    static_assert(CW <= 16 && DB > 2 * CW, "at most 64 dimensions per block, three or four per column");
    // bytes between the 16-element rows of the transpose scratch: 18 doubles / 20 floats -- 36 c / 20 c banks for column c are
    // sixteen different multiples of 4 (mod 64), so the 16 lanes of a ds_read_b128 service group never share a bank
    static constexpr int RED_STRIDE = sizeof(T) == 8 ? 18 * 8 : 20 * 4;
#ifndef PMF_LANE_XPOSE
#define PMF_LANE_XPOSE 2   // 0: swap folds + 16-lane transposes through LDS (round 3); 1: the whole reduction through an LDS image; 2: DPP butterfly + 2 swap levels
#endif
    // XPOSE (round 4; floats, every set in architectural registers): the gradient's 64-lane sums go through LDS WHOLE -- every lane
    // writes its KP lane-partials as KS 16-byte slots, every lane then adds up 16 lanes' worth of one slot-column and a quad of
    // lanes is combined by two DPP adds -- instead of through two levels of v_permlane32/16_swap folds and a 16-lane transpose.
    // Why: the probe of round 4 (scripts/probes/probe_lane.py) put the swap-fold reduction at 2 936 of a pass's 3 614 cycles on the
    // headline's item rows: a v_permlane*_swap whose operands come straight out of a multiply-add chain and whose result feeds an add
    // costs ~60 cycles of LATENCY (the 10 cycles of scripts/probes/valu_probe.hip are its issue rate back to back), and with 208 of
    // 254 registers holding the tile the 39 swaps of a pass run strictly one after the other.  The LDS version has one round trip.
    // k-vectors then live lane <-> dimension LINEARLY (dimension d in lane d): lane 4 q + e ends up with the sum of dimension 4 q + e.
    // Image: slot-column q (dimensions 4 q .. 4 q + 3) of lane l at slot XP_QS q + l + l / 16 -- one pad slot per 16 lanes and a
    // column stride = 4 (mod 16) make both the writes (a lane's own slot) and the reads (lane 4 q + p reads lanes 16 p .. 16 p + 15 of
    // column q: the 16 lanes of a ds_read_b128 service group hit 16 different bank quads) conflict-free.
    // (round 6: a PARTIAL LDS set may ride along -- LP_ > 0: its slots are read from LDS in the dots and in the butterfly's chains, everything else as without)
    static constexpr bool XPOSE = PMF_LANE_XPOSE && sizeof(T) == 4 && LL_ == 0 && (KP + WAVE - 1) / WAVE == 1;
    static constexpr int XP_QS = 68;
#ifndef PMF_LANE_COAL
#define PMF_LANE_COAL 1
#endif
    // COAL (round 4; with XPOSE): the register sets are fetched by COALESCED loads -- instruction i of a set reads the 64 consecutive
    // 16-byte slots 64 i .. 64 i + 63 of the row-major image [nonzero][W slots], i.e. W adjacent lanes share a factor row, as the
    // LDS-DMA chunks do -- straight into the tile's own registers (all sets of a row in flight at once), and each set is then turned
    // into "one nonzero per lane" by one trip through the XPOSE image in LDS (W ds_write_b128, W ds_read_b128).  The per-lane loads
    // this replaces named 64 different factor rows per instruction; the 13 instructions that touch a row's two lines were spread over
    // the whole gather of eight waves, far beyond what the 32 KB L1 keeps, so every 16-byte request was served by L2 again: the probe
    // of round 4 put the gather of a 1000-nonzero fp32 row at 26 k of its 62 k cycles.
    static constexpr bool COAL = PMF_LANE_COAL && XPOSE && LA_ == 0 && PMF_LANE_DIRECT;
    static_assert(!COAL || (STAGE_BYTES <= KS * XP_QS * 16 && KS == W), "a whole set's image fits the reduction's image");
    static constexpr int RED_PH = LP_ > 0 ? 2 : 1;        // the columns pass through the scratch in this many groups
    static constexpr int RED_COLS = RED_PH == 1 ? 16 : (CW + 1) / 2;
    static constexpr int TX = TX_;
    static_assert(TX_ == 0 || (LV_ == 1 && LA_ == 0 && LL_ == 0 && LP_ == 0 && NW_ == 1 && !SMALL_ && sizeof(T) == 8 && TX_ % 4 == 0 && TX_ <= WAVE),
                  "the LDS image of the tile: doubles, one register set, one wave per row");
    // (row-major [TX][KS + 1 slots]: the gradient reads ALONG a row -- conflict-free whatever the stride --, the dots and the image's writes
    // go DOWN a column, lane = row: a stride of 51 slots = 204 banks puts the 16 lanes of a ds_read_b128 service group on 16 different
    // bank quads.  53 040 bytes per wave at TX = 64: three rows per CU -- the 13-slot chunk images of the staging buffers, 54 064 bytes, made it two)
    static constexpr int TX_STRIDE = (KS + 1) * 16;
    static constexpr int TX_BYTES = TX_ > 0 ? TX_ * TX_STRIDE : 0;
    static constexpr int RED_BYTES = TX_ > 0 ? 0 : (XPOSE ? KS * XP_QS * 16 : 4 * RED_COLS * RED_STRIDE); // four 16-lane rows x the columns of a group
    // one chunk of the partial set: LP_ rows of W slots, rounded up to whole DMA instructions (64 lanes x 16 bytes) so that no
    // lane has to be masked off -- the lanes past the image fetch some row's slots into the padding
    static constexpr int PART_BYTES = LP_ > 0 ? (LP_ * W * 16 + 1023) / 1024 * 1024 : 0;
    static constexpr int AVEC_BYTES = (KP * (int)sizeof(T) + 15) / 16 * 16;
    static constexpr int WAVE_BYTES = NBUF * STAGE_BYTES + NCH * PART_BYTES + (ALIAS ? 0 : RED_BYTES) + TX_BYTES + AVEC_BYTES;
    static_assert(!ALIAS || RED_BYTES <= NBUF * STAGE_BYTES, "the reduction's scratch aliases the staging buffer(s): it must fit them (PMF_LANE_XPOSE = 1 on a SMALL instance does not)");
    // cross-wave scratch (NW > 1): two alternating sets of { NW x 64 NC doubles, NW scalars }
    static constexpr int XW_BYTES = NW_ > 1 ? NW_ * WAVE * NC * (int)sizeof(T) + 16 * ((NW_ * 8 + 15) / 16) : 0;
    static constexpr int SMEM_BYTES = NW * WAVE_BYTES + 2 * XW_BYTES + 16;
#ifndef PMF_LANE_SPOINT
#define PMF_LANE_SPOINT 1   // floats, all sets in registers: the point reaches the dots as SCALAR operands (set_point keeps it in a register,
                            // lane <-> dimension; eval() takes dimension c with one v_readlane and multiplies by the SGPR) instead of an LDS
                            // copy read back by broadcast -- no LDS round trip anywhere in the dots; same order of additions, same bits
#endif
    // (only where four or more sets share each v_readlane: with the one to three sets of the CG / TNCG instances the 52 readlanes of an
    // evaluation cost more than 13 broadcast reads -- C2 CG fp32 2.20 -> 2.58 ms, TNCG fp32 12.6 -> 13.5 with it)
    static constexpr bool SPOINT = PMF_LANE_SPOINT && sizeof(T) == 4 && NC == 1 && LL_ == 0 && LV_ >= 4;   // (LP_ > 0: the partial set's slots come from LDS, the point stays scalar)
#ifndef PMF_LANE_PIPE_MW
#define PMF_LANE_PIPE_MW 2   // sweep_rows' cross-row software pipeline (tickets two rows ahead, indices one) for MULTI-WAVE rows of this engine:
                             // 0 = never, 1 = always (rounds 3-4a), 2 = under PG only.  Measured on the C4 matrix (variant builds, same box):
                             // CG fp32 (eight waves per row) 20.20 -> 18.97 / 19.02 ms without it, B half 11.25 -> 10.25; CG fp64 (four waves)
                             // 37.11 -> 36.83; TNCG fp32 119.1 -> 118.5; PG(10) fp32 3.85 -> 3.94 ms for the dominant launch.  A workgroup-wide
                             // ticket is two barriers and an LDS round trip, and the next row's indices stay live across the whole solve.
#endif
    static constexpr bool PIPELINED = true;
    static constexpr int PIPE_MW = NW_ > 1 ? PMF_LANE_PIPE_MW : 1;   // (one-wave rows: always)
    static constexpr bool PARKS = false;
    static constexpr bool CACHED = true, MAY_CACHE = true, CACHED_GRAD = true;
    static constexpr int LS_BATCH = 1;
    static_assert(KS >= 13 && KP <= 2 * WAVE, "25 or 50 slots");
    static_assert(LV_ >= 1, "at least one set in architectural registers");

    T t[LV][KP];                          // the sets in architectural registers
    static constexpr int AW = (int)sizeof(T) / 4;   // 32-bit words per element
    typedef unsigned U4 __attribute__((ext_vector_type(4)));
    typedef unsigned UE __attribute__((ext_vector_type(sizeof(T) / 4)));
    UE tae[LA > 0 ? LA : 1][KP];   // the sets in accumulator registers, one AGPR (pair) per element (AGPR-class values: written by
                                   // acc_put or by a load straight into them, read by acc_get only).  (Quads per 16-byte slot, filled by
                                   // dwordx4 loads, cost the allocator 1.7 KB of scratch per lane: 128-bit tuples fragment the file.)
    T xcur;              // SPOINT: the current point, element of this lane's dimension (0 in lanes that hold none)
    T xr[LT];            // x_j of this lane's nonzeros
    unsigned idx_n[LT];  // column indices of the row whose tile is requested next (fetch_meta -> gather)
    T pv[LT], qv[LT];    // cached predictions p_j = F_j . x and q_j = F_j . d (solvers.hpp, cg_row_cached)
    const T* F;
    unsigned zero_row;
    int k, ldF;
    int lane, wid;
    int member;          // TM_: this workgroup's place in its team (0 otherwise)
    int tm_M;            // TM_: members of the team
    unsigned tm_seq;     // TM_: exchanges so far
    unsigned long long* tm_words;   // TM_: the team's exchange area (row_eval.hpp, team_sum)
    unsigned* tm_err;
    unsigned tm_spin;
    static constexpr unsigned TEAM_ROUND = 4;   // a member's share of a row is a whole number of these
    static_assert(!TM_ || (NW_ > 1 && sizeof(T) == 8), "lane teams: doubles, several waves per member");
    struct ElemOf {   // factor dimension held in element i of this lane (only meaningful where act[i])
        int d0;
        __device__ __forceinline__ int operator[](int i) const { return d0 + DB * i; }
    } elem;
    bool act[NC];
    bool holds_dim;      // this lane holds a dimension of the padded k-vector (elem.d0 < DB)
    unsigned nnz;        // nonzeros of the row held by THIS wave
    unsigned n_eval;
    unsigned char* stage;   // this wave's NBUF staging buffers
    unsigned char* part;    // this wave's partial LDS set (LP_ > 0): NCH chunk images of LP_ rows
    unsigned char* red;     // this wave's transpose scratch
    SA* avec;               // this wave's copy of the current point, as slots
    unsigned char* tximg;   // TX_ > 0: this wave's LDS image of the tile
    unsigned txoff[NC];     // TX_ > 0: byte offset of this lane's dimension i inside a row of the image (chunk base included)
    unsigned char* xw_base; // NW > 1: cross-wave scratch
    int xw_sel;
    unsigned* ticket_word;
    int pq_cap;
    T* pbuf;
    T* qbuf;
#ifdef PMF_PROBE
    unsigned* probe = nullptr;
#endif

    __device__ __forceinline__ void init(const TileGeom& geo, const T* F_, unsigned char* smem)
    {
        lane = lane_id();
        wid = NW > 1 ? (int)(threadIdx.x / WAVE) : 0;
        if constexpr (TM_) wid = (int)uniform((unsigned)wid);   // (a scalar to the compiler too: `if (wid == 0)` around the team exchange is a scalar branch)
        const int col = lane & 15, rr = lane >> 4;
        elem.d0 = XPOSE ? lane : col + CW * rr;
        const bool lane_on = XPOSE ? lane < DB : (col < CW && elem.d0 < DB);
        holds_dim = lane_on;
        F = F_;
        k = geo.k; ldF = geo.ldF; zero_row = geo.zero_row;
        unsigned char* p = smem + (size_t)wid * WAVE_BYTES;
        stage = p;
        part = p + NBUF * STAGE_BYTES;
        red = ALIAS ? p : p + NBUF * STAGE_BYTES + NCH * PART_BYTES;
        tximg = p + NBUF * STAGE_BYTES + NCH * PART_BYTES + (ALIAS ? 0 : RED_BYTES);
        avec = (SA*)(p + NBUF * STAGE_BYTES + NCH * PART_BYTES + (ALIAS ? 0 : RED_BYTES) + TX_BYTES);
        if constexpr (TX > 0) {
#pragma unroll
            for (int i = 0; i < NC; i++) {
                const int d = lane_on ? elem[i] : 0;
                txoff[i] = (unsigned)(d * (int)sizeof(T));
            }
        }
        xw_base = smem + (size_t)NW * WAVE_BYTES;
        xw_sel = 0;
        ticket_word = (unsigned*)(smem + (size_t)NW * WAVE_BYTES + 2 * XW_BYTES);
#pragma unroll
        for (int i = 0; i < NC; i++) act[i] = lane_on && elem[i] < k;
        pq_cap = 0x7fffffff;
        pbuf = (T*)(size_t)16; qbuf = (T*)(size_t)32;   // tags, never dereferenced
        n_eval = 0;
        nnz = 0;
        member = 0; tm_M = 1; tm_seq = 0; tm_words = nullptr; tm_err = nullptr; tm_spin = TEAM_SPIN_LIMIT;
    }
    __device__ __forceinline__ unsigned* ticket_slot() const { return ticket_word; }

    // solve_row: the point the row starts from
    __device__ __forceinline__ void start_point(const T* mrow, T (&x)[NC]) const { load_vec(mrow, x); }

    // ---- k-vectors: lane <-> dimension -------------------------------------------------------------------------------
    // Several wave-wide sums at once (the solvers' dot products come in groups: theta / beta / |g|^2; g.d / d.d; the three
    // sums of a line-search trial): two swap-fold levels put sum i into the lanes of 16-lane row (i & 1) * 2 + (i >> 1) --
    // three instructions per fold instead of one full reduction per value --, ONE 16-lane DPP reduction finishes all of
    // them, and a v_readlane pair per value hands them out.  29 instructions for four sums, 22 for two; 25 each one by one.
    static constexpr bool FUSED_SUMS = true;
    template <int N> __device__ __forceinline__ void rsum_n(T (&v)[N]) const
    {
        static_assert(N >= 1 && N <= 4, "up to four sums per reduction");
        T u;
        if constexpr (N == 1) u = swap_fold<16>(swap_fold<32>(v[0], v[0]), swap_fold<32>(v[0], v[0]));
        else if constexpr (N == 2) { const T w = swap_fold<32>(v[0], v[1]); u = swap_fold<16>(w, w); }
        else if constexpr (N == 3) u = swap_fold<16>(swap_fold<32>(v[0], v[1]), swap_fold<32>(v[2], v[2]));
        else u = swap_fold<16>(swap_fold<32>(v[0], v[1]), swap_fold<32>(v[2], v[3]));
        u = u + dpp_mov<0xB1>(u);
        u = u + dpp_mov<0x4E>(u);
        u = u + dpp_mov<0x141>(u);
        u = u + dpp_mov<0x140>(u);
        // rows: 0 holds v[0]; N >= 3: 1 holds v[2]; 2 holds v[1]; 3 holds v[3]   (N == 2: rows 0, 1 hold v[0], rows 2, 3 v[1])
        v[0] = uniform(u);
        if constexpr (N >= 2) v[1] = read_lane(u, 32);
        if constexpr (N >= 3) v[2] = read_lane(u, 16);
        if constexpr (N >= 4) v[3] = read_lane(u, 48);
    }
    template <class V> __device__ __forceinline__ V rsum(V x) const { return wave_sum(x); }
    template <class V> __device__ __forceinline__ V rmin(V x) const { return wave_min(x); }
    template <class V> __device__ __forceinline__ V rmax(V x) const { return wave_max(x); }
    __device__ __forceinline__ T dot(const T (&u)[NC], const T (&v)[NC]) const
    {
        T s = (T)0;
#pragma unroll
        for (int i = 0; i < NC; i++) s = act[i] ? fma_t(u[i], v[i], s) : s;
        return rsum(s);
    }
    __device__ __forceinline__ T nrm2(const T (&u)[NC]) const { return (T)d_sqrt((double)dot(u, u)); }
    __device__ __forceinline__ void load_vec(const T* p, T (&x)[NC]) const
    {
#pragma unroll
        for (int i = 0; i < NC; i++) x[i] = act[i] ? p[elem[i]] : (T)0;
    }
    __device__ __forceinline__ void store_vec(T* p, const T (&x)[NC]) const
    {
        if (wid == 0 && member == 0) {
#pragma unroll
            for (int i = 0; i < NC; i++)
                if (act[i]) p[elem[i]] = x[i];
        }
    }
    __device__ __forceinline__ void park(int, const T (&)[NC]) {}
    __device__ __forceinline__ void unpark(int, T (&)[NC]) {}

    // ---- gather --------------------------------------------------------------------------------------------------------
    // this wave's share [c0, c0 + mine) of a row of nnz_row nonzeros (a function of the row's length and NW alone)
    __device__ __forceinline__ void my_share(unsigned nnz_row, unsigned& c0, unsigned& mine) const
    {
        if constexpr (NW > 1) {
            const unsigned C = (nnz_row + NW - 1) / NW;
            c0 = (unsigned)wid * C;
            mine = c0 < nnz_row ? (nnz_row - c0 < C ? nnz_row - c0 : C) : 0u;
        } else {
            c0 = 0u; mine = nnz_row;
        }
    }
    __device__ __forceinline__ void fetch_meta(const unsigned* ind, unsigned nnz_row)
    {
        unsigned c0, mine;
        my_share(nnz_row, c0, mine);
#pragma unroll
        for (int s = 0; s < LT; s++) {
            const unsigned j = (unsigned)(WAVE * s + lane);
            idx_n[s] = j < mine ? ind[c0 + j] : zero_row;   // lanes past the end of the row fetch the all-zero row behind F
        }
    }
    __device__ __forceinline__ void begin_row(const unsigned* ind, const T* val, unsigned nnz_row)
    {
        fetch_meta(ind, nnz_row);
        gather(val, nnz_row);
    }

    static constexpr int chunk_start(int c) { return c * W < KS - W ? c * W : KS - W; }
    static constexpr int chunk_lo(int c) { return c == 0 ? 0 : chunk_start(c - 1) + W; }   // first slot the chunk is the first to bring
    static constexpr int slot_chunk(int q) { int c = 0; while (q >= chunk_start(c) + W) c++; return c; }   // the chunk slot q is read from

    // One chunk (slots [Q0, Q0 + W) of the 64 factor rows named by idx) -> staging buffer `buf`, as W LDS-DMA instructions.
    // Slot sigma = 64 i + lane of the row-major image [row][W slots] is row sigma / W, slot sigma % W.
    template <int Q0> __device__ __forceinline__ void dma_chunk(unsigned idx, int buf)
    {
        const unsigned rowbytes = (unsigned)ldF * (unsigned)sizeof(T);
        unsigned j4 = (unsigned)(lane / W) * 4u;            // byte address of the row's index for ds_bpermute
        unsigned q16 = (unsigned)(lane % W) * 16u;
        unsigned char* dst = stage + buf * STAGE_BYTES;
        static_for<0, W>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const unsigned c = (unsigned)__builtin_amdgcn_ds_bpermute((int)j4, (int)idx);
            const unsigned off = __umul24(c, rowbytes) + q16 + (unsigned)(Q0 * 16);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)F + (size_t)off),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            // next instruction: 64 slots further = 64 / W rows and 64 % W slots
            j4 += (unsigned)(WAVE / W) * 4u;
            q16 += (unsigned)(WAVE % W) * 16u;
            const bool wrap = q16 >= (unsigned)(W * 16);
            q16 = wrap ? q16 - (unsigned)(W * 16) : q16;
            j4 = wrap ? j4 + 4u : j4;
        });
    }
    // staged chunk -> this lane's slots [QLO, QHI) of register set S_ (the chunk starts at slot Q0)
    template <int S_, int Q0, int QLO, int QHI> __device__ __forceinline__ void read_chunk(int buf)
    {
        const SA* src = (const SA*)(stage + buf * STAGE_BYTES) + lane * W;
        static_for<QLO, QHI>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const SA v = src[q - Q0];
#pragma unroll
            for (int e = 0; e < SN; e++) {
                if constexpr (S_ < LV) t[S_][q * SN + e] = v.v[e];
                else if constexpr (sizeof(T) == 8) {
                    const unsigned long long b = __builtin_bit_cast(unsigned long long, v.v[e]);
                    tae[S_ - LV][q * SN + e][0] = acc_put((unsigned)b);
                    tae[S_ - LV][q * SN + e][1] = acc_put((unsigned)(b >> 32));
                } else tae[S_ - LV][q * SN + e][0] = acc_put(__builtin_bit_cast(unsigned, v.v[e]));
            }
        });
    }
    template <int OUTSTANDING_CHUNKS> __device__ __forceinline__ void wait_dma()
    {
        // LDS-DMA data is ordered for this wave's ds_reads by its own vmcnt (MI355X_MICROARCH.md, two waves per SIMD, item 7)
        constexpr int n = OUTSTANDING_CHUNKS * W;
        if constexpr (n == 0) __builtin_amdgcn_s_waitcnt(0x0f70);                                          // vmcnt(0)
        else if constexpr (n < 64) __builtin_amdgcn_s_waitcnt(0x0f70 | (n & 0xf) | ((n >> 4) << 14));      // vmcnt(n): the older chunks have landed
        wave_lds_fence();
    }
    __device__ __forceinline__ void gather(const T* val, unsigned nnz_row)
    {
        unsigned c0;
        my_share(nnz_row, c0, nnz);
        unsigned idx[LT];
#pragma unroll
        for (int s = 0; s < LT; s++) {
            const unsigned j = (unsigned)(WAVE * s + lane);
            idx[s] = idx_n[s];
            xr[s] = j < nnz ? val[c0 + j] : (T)0;
        }
        // Sets in architectural registers: every lane loads its own factor row, 16 bytes at a time, straight into the tile --
        // KS loads in flight per lane, no staging, no address arithmetic beyond the row's base.  (64 different rows per
        // instruction: the price is in the texture unit -- 64 tag look-ups per instruction -- which a row pays once; the lines
        // are the same 2-4 per row that a coalesced fetch would bring.)
        constexpr int S0 = DIRECT_V ? LV : 0;   // first set that goes through the staging buffers
        auto request_lds_sets = [&]() {
            static_for<0, LL * NCH>([&](auto wc) {
                constexpr int w = decltype(wc)::value;
                dma_chunk<chunk_start(w % NCH)>(idx[LR + w / NCH], w);
            });
            if constexpr (LP > 0) {
                static_for<0, NCH>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    dma_chunk_part<chunk_start(c)>(idx[LT - 1], c);
                });
            }
        };
        if constexpr (COAL) {
            const unsigned rowbytes = (unsigned)ldF * (unsigned)sizeof(T);
            // (the W (row, slot) pairs of a lane are the same for every row of the launch; derived from an opaque copy of the lane
            // number they are recomputed here -- five VALU instructions each -- instead of being kept in 26 registers across the
            // solver, i.e. in scratch)
            unsigned lane_here = (unsigned)lane;
            asm volatile("" : "+v"(lane_here));
            auto request_set = [&](auto sc) {
                constexpr int s2 = decltype(sc)::value;
                unsigned j4 = (lane_here / (unsigned)W) * 4u;            // byte address of the row's index for ds_bpermute
                unsigned q16 = (lane_here % (unsigned)W) * 16u;
                static_for<0, W>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const unsigned c = (unsigned)__builtin_amdgcn_ds_bpermute((int)j4, (int)idx[s2]);
                    const unsigned off = __umul24(c, rowbytes) + q16;
                    const typename Slot<T>::U v = *(const typename Slot<T>::U*)((const char*)F + (size_t)off);
#pragma unroll
                    for (int e = 0; e < SN; e++) t[s2][i * SN + e] = v.v[e];
                    j4 += (unsigned)(WAVE / W) * 4u;
                    q16 += (unsigned)(WAVE % W) * 16u;
                    const bool wrap = q16 >= (unsigned)(W * 16);
                    q16 = wrap ? q16 - (unsigned)(W * 16) : q16;
                    j4 = wrap ? j4 + 4u : j4;
                });
            };
            // image order -> one nonzero per lane through the image
            auto own_rows_of_image = [&](auto sc) {
                constexpr int s2 = decltype(sc)::value;
                const SA* rd = (const SA*)red + lane * W;
                static_for<0, W>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    const SA v = rd[q];
#pragma unroll
                    for (int e = 0; e < SN; e++) t[s2][q * SN + e] = v.v[e];
                });
                wave_lds_fence();
            };
            auto through_image = [&](auto sc) {
                constexpr int s2 = decltype(sc)::value;
                SA* wr = (SA*)red + lane;
                static_for<0, W>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    SA v;
#pragma unroll
                    for (int e = 0; e < SN; e++) v.v[e] = t[s2][i * SN + e];
                    wr[i * WAVE] = v;
                });
                wave_lds_fence();
                own_rows_of_image(sc);
            };
            if constexpr (LP > 0) request_lds_sets();   // (its DMA is in flight under the register sets' loads)
            static_for<0, LV>(request_set);
            static_for<0, LV>(through_image);
        } else if constexpr (TX > 0) {
            // TX: this lane's factor row, 16 bytes at a time, into free registers (the solver's state is dead between two rows) and
            // from there into the row-major image in LDS (lanes past the row's end fetch the zero row)
            const unsigned rowbytes = (unsigned)ldF * (unsigned)sizeof(T);
            const char* base = (const char*)F + (size_t)__umul24(idx[0], rowbytes);
            typename Slot<T>::U tmp[KS];
            static_for<0, KS>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                tmp[q] = *(const typename Slot<T>::U*)(base + q * 16);
            });
            wave_lds_fence();   // (the previous row's last reads of the image are older than these writes)
            if (TX == WAVE || lane < TX) {
                SA* row = (SA*)(tximg + lane * TX_STRIDE);
                static_for<0, KS>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    SA v;
#pragma unroll
                    for (int e = 0; e < SN; e++) v.v[e] = tmp[q].v[e];
                    row[q] = v;
                });
            }
            wave_lds_fence();
        } else if constexpr (DIRECT_V) {
            const unsigned rowbytes = (unsigned)ldF * (unsigned)sizeof(T);
            static_for<0, LV>([&](auto sc) {
                constexpr int s2 = decltype(sc)::value;
                const char* base = (const char*)F + (size_t)__umul24(idx[s2], rowbytes);
                static_for<0, KS>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    const typename Slot<T>::U v = *(const typename Slot<T>::U*)(base + q * 16);
#pragma unroll
                    for (int e = 0; e < SN; e++) t[s2][q * SN + e] = v.v[e];
                });
            });
        }
        // (without direct loads) the register sets' chunks pass through the staging buffers, NBUF in flight
        constexpr int NWORK = (LR - S0) * NCH;
        constexpr int DEPTH = NBUF < NWORK ? NBUF : NWORK;
        static_for<0, DEPTH>([&](auto wc) {
            constexpr int w = decltype(wc)::value;
            dma_chunk<chunk_start(w % NCH)>(idx[S0 + w / NCH], w % NBUF);
        });
        static_for<0, NWORK>([&](auto wc) {
            constexpr int w = decltype(wc)::value;
            constexpr int s = S0 + w / NCH, c = w % NCH;
            constexpr int later = NWORK - 1 - w;   // chunks requested after this one so far: min(later, DEPTH - 1)
            wait_dma<(later < DEPTH - 1 ? later : DEPTH - 1)>();
            read_chunk<s, chunk_start(c), chunk_lo(c), chunk_start(c) + W>(w % NBUF);
            if constexpr (w + DEPTH < NWORK || LL > 0) {
                // the reads of this buffer must have returned before the next DMA lands in it
                __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
                wave_lds_fence();
            }
            if constexpr (w + DEPTH < NWORK) {
                constexpr int w2 = w + DEPTH;
                dma_chunk<chunk_start(w2 % NCH)>(idx[S0 + w2 / NCH], w % NBUF);
            }
        });
        if constexpr (ALIAS) { __builtin_amdgcn_s_waitcnt(0xc07f); wave_lds_fence(); }   // the scratch of the reductions is this buffer
        // LDS sets: their chunks stay in the buffers (set u, chunk c in buffer u NCH + c)
        if constexpr (LL > 0 || LP > 0) {
            if constexpr (!(COAL && LP > 0)) request_lds_sets();
            wait_dma<0>();
        }
    }
    // the same for the partial set: LP rows, ceil(LP W / 64) instructions (ds_bpermute takes its lane index modulo 64: the lanes past
    // the image name rows of the set's first lanes)
    template <int Q0> __device__ __forceinline__ void dma_chunk_part(unsigned idx, int c)
    {
        const unsigned rowbytes = (unsigned)ldF * (unsigned)sizeof(T);
        unsigned j4 = (unsigned)(lane / W) * 4u;
        unsigned q16 = (unsigned)(lane % W) * 16u;
        unsigned char* dst = part + c * PART_BYTES;
        constexpr int NI = (LP * W + WAVE - 1) / WAVE;
        static_for<0, NI>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const unsigned col = (unsigned)__builtin_amdgcn_ds_bpermute((int)j4, (int)idx);
            const unsigned off = __umul24(col, rowbytes) + q16 + (unsigned)(Q0 * 16);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)F + (size_t)off),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            j4 += (unsigned)(WAVE / W) * 4u;
            q16 += (unsigned)(WAVE % W) * 16u;
            const bool wrap = q16 >= (unsigned)(W * 16);
            q16 = wrap ? q16 - (unsigned)(W * 16) : q16;
            j4 = wrap ? j4 + 4u : j4;
        });
    }

    __device__ __forceinline__ void set_point(const T (&x)[NC])
    {
        if constexpr (SPOINT) { xcur = act[0] ? x[0] : (T)0; return; }
        T* a = (T*)avec;
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < NC; i++)
            if (holds_dim) a[elem[i]] = act[i] ? x[i] : (T)0;   // (dimensions k .. KP - 1 of the padded row: zeros)
        wave_lds_fence();
    }

    // ---- tile access -----------------------------------------------------------------------------------------------------
    static __device__ __forceinline__ T acc_elem(const UE& w)
    {
        if constexpr (sizeof(T) == 8) {
            const unsigned lo = acc_get(w[0]), hi = acc_get(w[1]);
            return __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo);
        } else return __builtin_bit_cast(T, acc_get(w[0]));
    }
    // slot Q (SN elements) of this lane's nonzero in set S
    template <int S, int Q> __device__ __forceinline__ SA tile_slot() const
    {
        SA v;
        if constexpr (S < LV) {
#pragma unroll
            for (int e = 0; e < SN; e++) v.v[e] = t[S][Q * SN + e];
        } else if constexpr (S < LR) {
#pragma unroll
            for (int e = 0; e < SN; e++) v.v[e] = acc_elem(tae[S - LV][Q * SN + e]);
        } else {
            constexpr int c = slot_chunk(Q);
            if constexpr (S - LR < LL) v = *((const SA*)(stage + ((S - LR) * NCH + c) * STAGE_BYTES) + lane * W + (Q - chunk_start(c)));
            else v = *((const SA*)(part + c * PART_BYTES) + part_row(lane) * W + (Q - chunk_start(c)));   // (lanes >= LP: some row's finite data, coefficient 0)
        }
        return v;
    }
    // element C of this lane's nonzero in set S
    template <int S, int C> __device__ __forceinline__ T tile_elem() const
    {
        if constexpr (S < LV) return t[S][C];
        else if constexpr (S < LR) return acc_elem(tae[S - LV][C]);
        else {
            constexpr int q = C / SN, c = slot_chunk(q);
            if constexpr (S - LR < LL) return *((const T*)(stage + ((S - LR) * NCH + c) * STAGE_BYTES) + (lane * W + (q - chunk_start(c))) * SN + C % SN);
            else return *((const T*)(part + c * PART_BYTES) + (part_row(lane) * W + (q - chunk_start(c))) * SN + C % SN);
        }
    }

    // ---- the transposing reduction: KP lane-partials per lane -> the total of each dimension in the lane that holds it -----
    // The elements of the LDS sets that column COL needs (dimensions COL + CW r, r < 4, of block B), requested one column
    // ahead of their use; sched_barriers keep the compiler from requesting the whole tile at once (left alone it puts all 50
    // reads of a pass in flight and parks their 200 destination registers in AGPRs).
    static constexpr int LLX = LLT > 0 ? LLT : 1;
    template <int B, int COL> __device__ __forceinline__ void load_col(T (&tl)[4][LLX]) const
    {
        static_for<0, 4>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            constexpr int d = COL + CW * r;
            if constexpr (d < DB) {
                static_for<0, LLT>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    tl[r][u] = tile_elem<LR + u, DB * B + d>();
                });
            }
        });
    }
    // this lane's partial of dimension DB B + COL + CW R
    template <int B, int COL, int R> __device__ __forceinline__ T partial(const T (&coef)[LT], const T (&tl)[4][LLX]) const
    {
        constexpr int C = DB * B + COL + CW * R;
        T v = coef[0] * t[0][C];
        static_for<1, LR>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            v = fma_t(coef[s], tile_elem<s, C>(), v);
        });
        static_for<0, LLT>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            v = fma_t(coef[LR + u], tl[R][u], v);
        });
        return v;
    }
    // swap_fold for the transposing reduction.  Floats: through the LDS crossbar (ds_bpermute: the partner's copy of what this
    // lane collects) -- two selects and an add on the VALU (6 cycles) where v_permlane32_swap + add cost 12 (the swap alone ~10,
    // scripts/probes/valu_probe.hip); its latency is the other wave's issue time at two waves per SIMD.  The same sum, the same
    // bits.  Doubles (one wave per SIMD, nothing to hide a round trip behind) keep the swaps.
#ifndef PMF_LANE_PK
#define PMF_LANE_PK 1   // floats: packed multiply-adds (v_pk_fma_f32) in the transposing butterfly's chains
#endif
#ifndef PMF_LANE_PK_DOTS
#define PMF_LANE_PK_DOTS 1   // floats, the instances whose dots take the point as scalar operands (four sets and more): dimensions c, c + 1 in one v_pk_fma_f32 with
                             // an SGPR pair -- TWO partial sums per set (even and odd dimensions, added at the end), as the register engine's lane_dot has
                             // always done (reg_eval.hpp): PG(10) item rows 3.45 -> 3.15 ms.  (The instances of fewer sets read the point from LDS: packed there
                             // too, TNCG fp32 110.8 -> 109.9 ms, CG fp32 18.3 -> 18.0 for another order of their sums -- not kept.)
#endif
    template <int W_> __device__ __forceinline__ T fold(T a, T b) const
    {
        return swap_fold<W_>(a, b);   // (the folds through the LDS crossbar, ds_bpermute + two selects, lost everywhere: DESIGN.md 4.4)
    }
    // the two dimensions COL + CW R0 (kept by the lanes of the lower half-wave) and COL + CW (R0 + 2) (upper) of block B
    template <int B, int COL, int R0> __device__ __forceinline__ T level_a(const T (&coef)[LT], const T (&tl)[4][LLX]) const
    {
        static_assert(COL + CW * R0 < DB, "a dimension of the block");
        const T a = partial<B, COL, R0>(coef, tl);
        if constexpr (COL + CW * (R0 + 2) < DB) return fold<32>(a, partial<B, COL, R0 + 2>(coef, tl));
        else return fold<32>(a, a);   // (the upper half-wave's result belongs to no dimension)
    }
    template <int B> __device__ __forceinline__ T reduce_block(const T (&coef)[LT])
    {
        if constexpr (XPOSE && PMF_LANE_XPOSE == 2) {
            // (mode 2) No LDS and three swaps: inside every 16-lane row the KP lane-partials are summed by reg_eval.hpp's TRANSPOSING
            // BUTTERFLY, sixteen at a time -- four levels of "keep the register your lane class collects, add the partner lane's copy of
            // it" (two selects and one DPP add per fold, 15 folds per batch) leave the row's sum of partial 16 b + g in lane g -- and
            // the four rows' sums of the (up to) four batches meet in two swap folds: row r ends up with batch r, i.e. lane l with the
            // total of dimension l.  ~160 VALU instructions for 52 dimensions, dependent chains of six operations.
            static_assert(B == 0 && KP <= WAVE, "one element per lane");
            const bool c8 = (lane & 8) != 0, c4 = (lane & 4) != 0, c2 = (lane & 2) != 0, c1 = (lane & 1) != 0;
            constexpr int NBATCH = (KP + 15) / 16;
            T rowsum[4] = { (T)0, (T)0, (T)0, (T)0 };
            static_for<0, NBATCH>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                constexpr int N = KP - 16 * b < 16 ? KP - 16 * b : 16;        // partials of this batch
                T q[8], r[4], s2[2];
                constexpr int N8 = N < 8 ? N : 8;
                {
                    // the lane-partials of eight dimensions at a time, SET-major: eight independent multiply-add chains side by side
                    // (column by column the chains were two wide: a dependent v_fmac every other instruction)
                    static_for<0, 2>([&](auto hc) {
                        constexpr int h = decltype(hc)::value;
                        T u[8];
                        // chain j: dimension 16 b + 4 h + j / 2 + 8 (j % 2)   (the two inputs of fold 4 h + j / 2)
                        static_assert(LLT == 0 || (sizeof(T) == 4 && PMF_LANE_PK && N % 4 == 0 && LV >= 2 && LV == LR), "a partial LDS set rides in the packed chains only");
                        if constexpr (sizeof(T) == 4 && PMF_LANE_PK && N % 2 == 0 && LV >= 2) {   // (one set: the lone multiply is contracted with the fold's add by the compiler -- kept as it is, bit for bit)
                            // floats: the chains of two NEIGHBOURING dimensions in one v_pk_fma_f32 (the tile's elements C, C + 1 sit in neighbouring
                            // registers: they arrive four to a 16-byte load) -- the same multiply-adds in the same order, half the instructions
                            typedef float v2f __attribute__((ext_vector_type(2)));
                            v2f u2[4];
                            static_for<0, LV>([&](auto sc) {
                                constexpr int s_ = decltype(sc)::value;
                                static_for<0, 4>([&](auto pc) {
                                    constexpr int p = decltype(pc)::value;
                                    constexpr int d = 4 * h + 2 * (p % 2) + 8 * (p / 2);      // pair p: dimensions d, d + 1 = chains j0, j0 + 2, j0 = 4 (p % 2) + p / 2
                                    if constexpr (d < N) {
                                        constexpr int C = 16 * b + d;
                                        const v2f tt = { (float)t[s_][C], (float)t[s_][C + 1] };
                                        const v2f cc = { (float)coef[s_], (float)coef[s_] };
                                        if constexpr (s_ == 0) u2[p] = cc * tt;
                                        else u2[p] = __builtin_elementwise_fma(cc, tt, u2[p]);
                                    }
                                });
                            });
                            if constexpr (LLT > 0) {
                                // the partial LDS set's elements of these dimensions: slots 4 b + h (pairs 0, 1) and 4 b + h + 2 (pairs 2, 3), last in every chain
                                static_for<0, 2>([&](auto gc2) {
                                    constexpr int g2 = decltype(gc2)::value;
                                    if constexpr (4 * h + 8 * g2 < N) {
                                        const SA ls = tile_slot<LR, 4 * b + h + 2 * g2>();
                                        const v2f cc = { (float)coef[LR], (float)coef[LR] };
                                        u2[2 * g2] = __builtin_elementwise_fma(cc, v2f{ (float)ls.v[0], (float)ls.v[1] }, u2[2 * g2]);
                                        u2[2 * g2 + 1] = __builtin_elementwise_fma(cc, v2f{ (float)ls.v[2], (float)ls.v[3] }, u2[2 * g2 + 1]);
                                    }
                                });
                            }
                            static_for<0, 4>([&](auto pc) {
                                constexpr int p = decltype(pc)::value;
                                constexpr int d = 4 * h + 2 * (p % 2) + 8 * (p / 2);
                                if constexpr (d < N) { u[4 * (p % 2) + p / 2] = (T)u2[p].x; u[4 * (p % 2) + p / 2 + 2] = (T)u2[p].y; }
                            });
                        } else
                        static_for<0, LV>([&](auto sc) {
                            constexpr int s_ = decltype(sc)::value;
                            static_for<0, 8>([&](auto jc) {
                                constexpr int j = decltype(jc)::value;
                                constexpr int d = 4 * h + j / 2 + 8 * (j % 2);
                                if constexpr (d < N) {
                                    constexpr int C = 16 * b + d;
                                    if constexpr (s_ == 0) u[j] = coef[0] * t[0][C];
                                    else u[j] = fma_t(coef[s_], t[s_][C], u[j]);
                                }
                            });
                        });
                        if constexpr (sizeof(T) == 4 && N == 16) {                  // lane ^ 8 (row_ror:8), four folds in place
                            fold4_ror8_inplace(u[0], u[2], u[4], u[6], u[1], u[3], u[5], u[7]);
                            q[4 * h] = u[0]; q[4 * h + 1] = u[2]; q[4 * h + 2] = u[4]; q[4 * h + 3] = u[6];
                        } else
                        static_for<0, 4>([&](auto ic) {                             // lane ^ 8 (row_ror:8)
                            constexpr int i = 4 * h + decltype(ic)::value;
                            constexpr int j = 2 * decltype(ic)::value;
                            if constexpr (i + 8 < N) q[i] = fold_pair<0x128>(c8, u[j], u[j + 1]);
                            else if constexpr (i < N) q[i] = fold_one<0x128>(u[j]);
                            else q[i] = (T)0;
                        });
                    });
                    if constexpr (sizeof(T) == 4 && N8 == 8) {                      // lane ^ 7 (row_half_mirror), four folds in place
                        fold4_mirror_inplace(q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]);
                        r[0] = q[0]; r[1] = q[1]; r[2] = q[2]; r[3] = q[3];
                    } else
                    fold_level<0x141, 4, N8>(c4, q, r);                             // lane ^ 7 (row_half_mirror)
                }
                constexpr int N4 = N8 < 4 ? N8 : 4;
                fold_level<0x4E, 2, N4>(c2, r, s2);                             // lane ^ 2
                constexpr int N2 = N4 < 2 ? N4 : 2;
                if constexpr (N2 == 2) rowsum[b] = fold_pair<0xB1>(c1, s2[0], s2[1]);   // lane ^ 1
                else rowsum[b] = fold_one<0xB1>(s2[0]);
#ifdef PMF_PROBE
                asm volatile("" : "+v"(rowsum[b]));
                PMF_STAMP(*this, 5 + b);
#endif
            });
            // rows 0 | 2 collect batch 0 | 2, rows 1 | 3 batch 1 | 3
            const T x = swap_fold<32>(rowsum[0], rowsum[2]), y = swap_fold<32>(rowsum[1], rowsum[3]);
            return swap_fold<16>(x, y);
        }
        if constexpr (XPOSE) {
            static_assert(B == 0 && SN == 4, "floats, one element per lane");
            SA* img = (SA*)red + (lane + (lane >> 4));
            static_for<0, KS>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                SA v;
                static_for<0, SN>([&](auto ec) {
                    constexpr int C = q * SN + decltype(ec)::value;
                    T u = coef[0] * t[0][C];
                    static_for<1, LV>([&](auto sc) { u = fma_t(coef[decltype(sc)::value], t[decltype(sc)::value][C], u); });
                    v.v[C % SN] = u;
                });
                img[q * XP_QS] = v;
            });
            wave_lds_fence();
            // lane 4 q + p: lanes 16 p .. 16 p + 15 of slot-column q, in lane order; then the four quarters (p0 + p1) + (p2 + p3)
            // (lanes past the last slot-column hold no dimension: they read column q - 4 -- valid memory, and 4 x 68 slots further down
            // is the same bank quad as their own column would be, so their service group stays conflict-free)
            static_assert(KS >= 12 && KS <= 16, "slot-columns of the lanes past 4 KS");
            const int qr = (lane >> 2) < KS ? (lane >> 2) : (lane >> 2) - 4, pr = lane & 3;
            const SA* rd = (const SA*)red + (qr * XP_QS + 17 * pr);
            SA a = rd[0];
#pragma unroll
            for (int i = 1; i < 16; i++) {
                const SA b = rd[i];
#pragma unroll
                for (int e = 0; e < SN; e++) a.v[e] += b.v[e];
            }
#pragma unroll
            for (int e = 0; e < SN; e++) a.v[e] = a.v[e] + dpp_mov<0xB1>(a.v[e]);   // quad_perm [1, 0, 3, 2]
#pragma unroll
            for (int e = 0; e < SN; e++) a.v[e] = a.v[e] + dpp_mov<0x4E>(a.v[e]);   // quad_perm [2, 3, 0, 1]
            const T r01 = (pr & 1) ? a.v[1] : a.v[0], r23 = (pr & 1) ? a.v[3] : a.v[2];
            wave_lds_fence();                              // (the image is rewritten by the next pass)
            return (pr & 2) ? r23 : r01;                   // lane 4 q + p: dimension 4 q + p
        }
        const int R = lane >> 4, p = lane & 15;
        unsigned char* wr = red + R * (RED_COLS * RED_STRIDE) + p * (int)sizeof(T);
        T tl[2][4][LLX];
        T result = (T)0;
        load_col<B, 0>(tl[0]);
        static_for<0, CW>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if constexpr (c + 1 < CW) load_col<B, c + 1>(tl[(c + 1) & 1]);
            if constexpr (LLT > 0) pin_here();   // (no LDS set: nothing to hold back, and the columns' dependent chains -- four multiply-adds, swap,
                                                 // add, swap, add -- are better interleaved than run one after the other)
            const T x = level_a<B, c, 0>(coef, tl[c & 1]);         // rows 0 | 2
            T o;
            if constexpr (c + CW < DB) o = fold<16>(x, level_a<B, c, 1>(coef, tl[c & 1]));   // rows 1 | 3
            else o = fold<16>(x, x);
            *(T*)(wr + (c % RED_COLS) * RED_STRIDE) = o;             // lane (R, p): dimension c + CW R, summed over the four lanes (., p)
            if constexpr (LLT > 0) pin_here();
            // the last column of a group: lane (R, c') adds up column c' of the group over the 16 lanes of its row
            if constexpr ((c + 1) % RED_COLS == 0 || c + 1 == CW) {
                constexpr int c_first = c / RED_COLS * RED_COLS;
                wave_lds_fence();
                const int pc = p >= c_first ? p - c_first : 0;
                const SA* rd = (const SA*)(red + R * (RED_COLS * RED_STRIDE) + (pc < RED_COLS ? pc : 0) * RED_STRIDE);
                SA v[16 / SN];
#pragma unroll
                for (int i = 0; i < 16 / SN; i++) v[i] = rd[i];
                T sum = v[0].v[0];
#pragma unroll
                for (int i = 1; i < 16; i++) sum += v[i / SN].v[i % SN];
                result = (p >= c_first && p <= c) ? sum : result;
                wave_lds_fence();
            }
        });
        return result;                                  // lane (R, c): dimension c + CW R (lanes with c >= CW: nothing)
    }

    // TX: tot_i = sum_j coef_j T[j][dimension i of this lane], j in the row's own order (ONES: every coefficient 1).  Four nonzeros per
    // block, the next block's eight elements requested before this block's multiply-adds; blocks past the row's end are skipped (wave-uniform).
    template <bool ONES> __device__ __forceinline__ void tx_axpy(T coef, T (&tot)[NC]) const
    {
        constexpr int JB = 4, NB = TX / JB;
        T acc[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) acc[i] = (T)0;
        T buf[2][JB][NC];
        auto load = [&](auto bc, T (&b_)[JB][NC]) {
            constexpr int j0 = decltype(bc)::value * JB;
#pragma unroll
            for (int u = 0; u < JB; u++) {
#pragma unroll
                for (int i = 0; i < NC; i++) b_[u][i] = *(const T*)(tximg + txoff[i] + (j0 + u) * TX_STRIDE);
            }
        };
        const unsigned n_here = uniform(nnz);   // (wave-uniform, and said so: the blocks below are scalar branches)
        load(std::integral_constant<int, 0>{}, buf[0]);
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            if (n_here > (unsigned)(JB * b)) {
                if constexpr (b + 1 < NB) load(std::integral_constant<int, b + 1>{}, buf[(b + 1) & 1]);
#pragma unroll
                for (int u = 0; u < JB; u++) {
                    if constexpr (ONES) {
#pragma unroll
                        for (int i = 0; i < NC; i++) acc[i] += buf[b & 1][u][i];
                    } else {
                        const T c = read_lane(coef, JB * b + u);
#pragma unroll
                        for (int i = 0; i < NC; i++) acc[i] = fma_t(c, buf[b & 1][u][i], acc[i]);
                    }
                }
            }
        });
#pragma unroll
        for (int i = 0; i < NC; i++) tot[i] = act[i] ? acc[i] : (T)0;
    }

    // NW > 1: add up the NW waves' results (fixed order; every wave ends with the same bits)
    __device__ __forceinline__ void combine_waves(T (&tot)[NC], double& lsum, bool vec = true)
    {
        if constexpr (NW > 1) {
            T* xv = (T*)(xw_base + xw_sel * XW_BYTES);
            double* xl = (double*)(xw_base + xw_sel * XW_BYTES + NW * WAVE * NC * (int)sizeof(T));
            xw_sel ^= 1;
            if (vec) {
#pragma unroll
                for (int i = 0; i < NC; i++) xv[(wid * NC + i) * WAVE + lane] = tot[i];
            }
            if (TM_ || lane == 0) xl[wid] = lsum;   // (TM_: no lane-divergent region between here and the team exchange, see team_sum)
            __syncthreads();
            double lp[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) lp[w] = xl[w];
            lsum = 0.0;
#pragma unroll
            for (int w = 0; w < NW; w++) lsum += lp[w];
            if (vec) {
                T part[NW][NC];
#pragma unroll
                for (int w = 0; w < NW; w++) {
#pragma unroll
                    for (int i = 0; i < NC; i++) part[w][i] = xv[(w * NC + i) * WAVE + lane];
                }
#pragma unroll
                for (int i = 0; i < NC; i++) {
                    T s = part[0][i];
#pragma unroll
                    for (int w = 1; w < NW; w++) s += part[w][i];
                    tot[i] = s;
                }
            }
            if constexpr (TM_) team_exchange(tot, lsum, vec);
        }
    }
    // TM_, every wave of the member, after the member's own waves have been added up (tot / lsum identical in all of them): the member's sums
    // cross the team (row_eval.hpp, team_sum: tagged granules, member order); the totals reach the other waves through the cross-wave
    // scratch -- the set the NEXT combine writes -- between two barriers.
    __device__ __forceinline__ void team_exchange(T (&tot)[NC], double& lsum, bool vec)
    {
        tm_seq++;
        T* xv = (T*)(xw_base + xw_sel * XW_BYTES);
        double* xl = (double*)(xw_base + xw_sel * XW_BYTES + NW * WAVE * NC * (int)sizeof(T));
        if (wid == 0) {
            TeamVals<NC> x;
#pragma unroll
            for (int i = 0; i < NC; i++) { x.idx[i] = act[i] ? elem[i] : 0; x.valid[i] = vec && act[i]; x.v[i] = (double)tot[i]; }
            x.s = lsum;
            x.ok = true;
            x = team_sum_call<NC>(tm_words, tm_M, member, tm_seq, tm_err, tm_spin, lane, x);   // (out of line: see row_eval.hpp)
#pragma unroll
            for (int i = 0; i < NC; i++) xv[i * WAVE + lane] = (T)x.v[i];
            xl[0] = x.s;
        }
        __syncthreads();
        lsum = xl[0];
        if (vec) {
#pragma unroll
            for (int i = 0; i < NC; i++) tot[i] = act[i] ? xv[i * WAVE + lane] : (T)0;
        }
        __syncthreads();
    }

    // Same contract as RegEval::eval
    template <bool WANT_F, bool WANT_G, bool FROM_CACHE = false> __device__ __forceinline__ double eval(T sgn, T (&acc)[NC], T* store = nullptr)
    {
        n_eval++;
        PMF_STAMP(*this, 0);
        T pred[LT];
        if constexpr (FROM_CACHE) {
#pragma unroll
            for (int s = 0; s < LT; s++) pred[s] = pv[s];
        } else if constexpr (SPOINT) {
            // dimension c of the point sits in lane (c % CW) + 16 (c / CW) (XPOSE: in lane c): one v_readlane, then a scalar operand of LT multiply-adds
            // (read in groups of GS ahead of their use: a v_readlane's SGPR needs two wait states before a VALU may read it)
            constexpr int GS = 4;
            constexpr bool PKD = sizeof(T) == 4 && PMF_LANE_PK_DOTS && KP % GS == 0;
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f pred2[LT];   // PKD: even / odd dimensions' partial dot products, two multiply-adds per v_pk_fma_f32
            static_assert(LLT == 0 || (PKD && GS == SN && LV == LR), "a partial LDS set under scalar-operand dots: one 16-byte slot per group of dimensions");
            SA pslot[2];     // the partial LDS set's slot of this group of dimensions and of the next (requested one group ahead)
            if constexpr (LLT > 0) pslot[0] = tile_slot<LR, 0>();
            static_for<0, (KP + GS - 1) / GS>([&](auto gc) {
                constexpr int c0 = decltype(gc)::value * GS;
                if constexpr (LLT > 0 && c0 / GS + 1 < KS) pslot[(c0 / GS + 1) & 1] = tile_slot<LR, c0 / GS + 1>();
                T ac[GS];
                static_for<0, GS>([&](auto ic) {
                    constexpr int c = c0 + decltype(ic)::value;
                    if constexpr (c < KP) {
                        ac[c - c0] = read_lane(xcur, XPOSE ? c : (c % CW) + 16 * (c / CW));
                        asm volatile("" : "+s"(ac[c - c0]));   // here, not sunk next to its use
                    }
                });
                if constexpr (PKD) {
                    static_for<0, GS / 2>([&](auto hc) {
                        constexpr int c = c0 + 2 * decltype(hc)::value;
                        const v2f a2 = { (float)ac[c - c0], (float)ac[c - c0 + 1] };
                        static_for<0, LT>([&](auto sc) {
                            constexpr int s2 = decltype(sc)::value;
                            v2f t2;
                            if constexpr (s2 < LV) t2 = v2f{ (float)t[s2][c], (float)t[s2][c + 1] };
                            else t2 = v2f{ (float)pslot[(c0 / GS) & 1].v[c - c0], (float)pslot[(c0 / GS) & 1].v[c - c0 + 1] };
                            if constexpr (c == 0) pred2[s2] = t2 * a2;
                            else pred2[s2] = __builtin_elementwise_fma(t2, a2, pred2[s2]);
                        });
                    });
                } else
                static_for<0, GS>([&](auto ic) {
                    constexpr int c = c0 + decltype(ic)::value;
                    if constexpr (c < KP) {
                        static_for<0, LV>([&](auto sc) {
                            constexpr int s2 = decltype(sc)::value;
                            if constexpr (c == 0) pred[s2] = t[s2][c] * ac[c - c0];
                            else pred[s2] = fma_t(t[s2][c], ac[c - c0], pred[s2]);
                        });
                    }
                });
            });
            if constexpr (PKD) {
#pragma unroll
                for (int s2 = 0; s2 < LT; s2++) pred[s2] = (T)(pred2[s2].x + pred2[s2].y);
            }
        } else if constexpr (TX > 0) {
            // TX: lane j's own row of the LDS image against the point (a broadcast read), slots in groups of GQ, the next group requested
            // before the current group's multiply-adds; the plain left-to-right chain of KP multiply-adds of the other instances
            auto dots = [&](T (&pred)[LT]) {
                constexpr int GQ = 5, NG = (KS + GQ - 1) / GQ;
                const SA* row = (const SA*)(tximg + ((TX == WAVE || lane < TX) ? lane : 0) * TX_STRIDE);   // (lanes past the image: some row's finite data, coefficient 0)
                SA av[2][GQ], tv[2][GQ];
                auto load_group = [&](auto gc, SA (&a_)[GQ], SA (&t_)[GQ]) {
                    constexpr int g = decltype(gc)::value;
                    static_for<0, GQ>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, q = g * GQ + i;
                        if constexpr (q < KS) { a_[i] = avec[q]; t_[i] = row[q]; }
                    });
                };
                load_group(std::integral_constant<int, 0>{}, av[0], tv[0]);
                static_for<0, NG>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    if constexpr (g + 1 < NG) load_group(std::integral_constant<int, g + 1>{}, av[(g + 1) & 1], tv[(g + 1) & 1]);
                    pin_here();
                    static_for<0, GQ>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, q = g * GQ + i;
                        if constexpr (q < KS) {
                            if constexpr (q == 0) pred[0] = tv[g & 1][i].v[0] * av[g & 1][i].v[0];
                            else pred[0] = fma_t(tv[g & 1][i].v[0], av[g & 1][i].v[0], pred[0]);
#pragma unroll
                            for (int e = 1; e < SN; e++) pred[0] = fma_t(tv[g & 1][i].v[e], av[g & 1][i].v[e], pred[0]);
                        }
                    });
                    asm volatile("" : "+v"(pred[0]));
                    pin_here();
                });
            };
#ifdef PMF_DUP_DOTS
            { T p2[LT]; dots(p2); asm volatile("" :: "v"(p2[0])); }
#endif
            dots(pred);
        } else {
            auto dots = [&](T (&pred)[LT]) {
            // slots in groups of GQ: the point (a broadcast read) and the LDS sets' slots of the NEXT group are requested
            // before the current group's multiply-adds
            constexpr int GQ = PMF_LANE_GQ(LA_, LP_), NG = (KS + GQ - 1) / GQ;
            SA av[2][GQ], tl[2][LLX][GQ];
            auto load_group = [&](auto gc, SA (&a_)[GQ], SA (&t_)[LLX][GQ]) {
                constexpr int g = decltype(gc)::value;
                static_for<0, GQ>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, q = g * GQ + i;
                    if constexpr (q < KS) {
                        a_[i] = avec[q];                    // the same address in every lane
                        static_for<0, LLT>([&](auto uc) {
                            constexpr int u = decltype(uc)::value;
                            t_[u][i] = tile_slot<LR + u, q>();
                        });
                    }
                });
            };
            load_group(std::integral_constant<int, 0>{}, av[0], tl[0]);
            static_for<0, NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if constexpr (g + 1 < NG) load_group(std::integral_constant<int, g + 1>{}, av[(g + 1) & 1], tl[(g + 1) & 1]);
                pin_here();
                static_for<0, GQ>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, q = g * GQ + i;
                    if constexpr (q < KS) {
                        static_for<0, LT>([&](auto sc) {
                            constexpr int s2 = decltype(sc)::value;
                            SA tv;
                            if constexpr (s2 < LR) tv = tile_slot<s2, q>();
                            else tv = tl[g & 1][s2 - LR][i];
                            if constexpr (q == 0) pred[s2] = tv.v[0] * av[g & 1][i].v[0];
                            else pred[s2] = fma_t(tv.v[0], av[g & 1][i].v[0], pred[s2]);
#pragma unroll
                            for (int e = 1; e < SN; e++) pred[s2] = fma_t(tv.v[e], av[g & 1][i].v[e], pred[s2]);
                        });
                    }
                });
                // (the multiply-adds are no memory operations: what ties them to this side of the line is the value they produce)
#pragma unroll
                for (int s2 = 0; s2 < LT; s2++) asm volatile("" : "+v"(pred[s2]));
                pin_here();
            });
            };
#ifdef PMF_DUP_DOTS   // development: the dots made twice (same results): the difference to the plain build is what they cost
            { T p2[LT]; dots(p2); for (int s2 = 0; s2 < LT; s2++) asm volatile("" :: "v"(p2[s2])); }
#endif
            dots(pred);
        }
        PMF_STAMP(*this, 1);
        if (store == pbuf) {
#pragma unroll
            for (int s = 0; s < LT; s++) pv[s] = pred[s];
        } else if (store == qbuf) {
#pragma unroll
            for (int s = 0; s < LT; s++) qv[s] = pred[s];
        }
        double lpart = 0.0;
        T coef[LT];
#ifdef PMF_DUP_COEF   // development: logarithms and divisions made twice
        {
            double l2nd = 0.0;
            T c2[LT];
#pragma unroll
            for (int s = 0; s < LT; s++) {
                T pr2 = pred[s];
                asm volatile("" : "+v"(pr2));
                const bool on = (unsigned)(WAVE * s + lane) < nnz;
                if constexpr (WANT_F) l2nd += on ? (double)xr[s] * d_log((double)pr2) : 0.0;
                if constexpr (WANT_G) { c2[s] = on ? coef_div(sgn * xr[s], pr2) : (T)0; asm volatile("" :: "v"(c2[s])); }
            }
            asm volatile("" :: "v"(l2nd));
        }
#endif
#pragma unroll
        for (int s = 0; s < LT; s++) {
            const bool on = (unsigned)(WAVE * s + lane) < nnz;
            if constexpr (WANT_F) lpart += on ? (double)xr[s] * d_log((double)pred[s]) : 0.0;
            if constexpr (WANT_G) coef[s] = on ? coef_div(sgn * xr[s], pred[s]) : (T)0;
        }
        T tot[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) tot[i] = (T)0;
        PMF_STAMP(*this, 2);
        if constexpr (WANT_G && TX > 0) {
#ifdef PMF_DUP_RED
            { T t2[NC]; tx_axpy<false>(coef[0], t2); asm volatile("" :: "v"(t2[0])); }
#endif
            tx_axpy<false>(coef[0], tot);
        } else if constexpr (WANT_G) {
#ifdef PMF_DUP_RED   // development: the transposing reduction (with its partial products) made twice
            static_for<0, NC>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                T r = reduce_block<b>(coef);
                asm volatile("" :: "v"(r));
            });
#endif
            static_for<0, NC>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                const T r = reduce_block<b>(coef);
                tot[b] = act[b] ? r : (T)0;
            });
        }
        PMF_STAMP(*this, 3);
        if constexpr (NW > 1 && !WANT_F && !WANT_G) {
            return 0.0;
        } else if constexpr (NW > 1) {
            double lsum = 0.0;
            if constexpr (WANT_F) lsum = wave_sum(lpart);
            combine_waves(tot, lsum, WANT_G);
            PMF_STAMP(*this, 4);
            if constexpr (WANT_G) {
#pragma unroll
                for (int i = 0; i < NC; i++) acc[i] += tot[i];
            }
            return lsum;
        } else {
            if constexpr (WANT_G) {
#pragma unroll
                for (int i = 0; i < NC; i++) acc[i] += tot[i];
            }
            if constexpr (WANT_F) return wave_sum(lpart);
            else return 0.0;
        }
    }

    // sum_j x_j log(p_j + alpha q_j) from the cached predictions (see RegEval::logsum_cached)
    __device__ __forceinline__ double logsum_cached(T alpha, bool& trusted)
    {
        double lpart = 0.0;
        bool bad = false;
#pragma unroll
        for (int s = 0; s < LT; s++) {
            const bool on = (unsigned)(WAVE * s + lane) < nnz;
            const T pred = fma_t(alpha, qv[s], pv[s]);
            bad = bad || (on && !(pred > pv[s] * (T)1e-4));
            lpart += on ? (double)xr[s] * d_log((double)pred) : 0.0;
        }
        trusted = __builtin_amdgcn_ballot_w64(bad) == 0;
        double l = wave_sum(lpart);
        if constexpr (NW > 1) {
            if (!trusted) l = __builtin_nan("");
            T none[NC];
#pragma unroll
            for (int i = 0; i < NC; i++) none[i] = (T)0;
            combine_waves(none, l, false);
            trusted = !(l != l);
        }
        return l;
    }
    // the same, un-reduced: this lane's share of sum_j x_j log(p_j + alpha q_j), and whether any of its predictions cancelled
    __device__ __forceinline__ double logsum_cached_lane(T alpha, bool& bad) const
    {
        double lpart = 0.0;
        bad = false;
#pragma unroll
        for (int s = 0; s < LT; s++) {
            const bool on = (unsigned)(WAVE * s + lane) < nnz;
            const T pred = fma_t(alpha, qv[s], pv[s]);
            bad = bad || (on && !(pred > pv[s] * (T)1e-4));
            lpart += on ? (double)xr[s] * d_log((double)pred) : 0.0;
        }
        return lpart;
    }
    // NW > 1: a wave-uniform scalar summed over the row's waves (NaN in any wave poisons the sum: every wave takes the same branch)
    __device__ __forceinline__ double combine_scalar(double l)
    {
        if constexpr (NW > 1) {
            T none[NC];
#pragma unroll
            for (int i = 0; i < NC; i++) none[i] = (T)0;
            combine_waves(none, l, false);
        }
        return l;
    }
    __device__ __forceinline__ void logsum_cached_batch(T alpha, T, double (&ls)[LS_BATCH], bool (&trusted)[LS_BATCH])
    {
        ls[0] = logsum_cached(alpha, trusted[0]);
    }
    __device__ __forceinline__ void advance_cached(T alpha)
    {
#pragma unroll
        for (int s = 0; s < LT; s++) pv[s] = fma_t(alpha, qv[s], pv[s]);
    }

    // acc_c += sum_j F[ind_j, c]  (adjustment_Bsum's gather pass, ref: src/poismf.c:108-110)
    __device__ __forceinline__ void tile_colsum(T (&acc)[NC])
    {
        T one[LT];
#pragma unroll
        for (int s = 0; s < LT; s++) one[s] = (T)1;   // lanes past the row's end hold the zero row
        if constexpr (LP > 0) one[LT - 1] = lane < LP ? (T)1 : (T)0;   // (the partial set's lanes >= LP alias rows of its first lanes)
        T tot[NC];
        if constexpr (TX > 0) tx_axpy<true>((T)1, tot);   // (rows past the end of the row are the zero row)
        else static_for<0, NC>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            const T r = reduce_block<b>(one);
            tot[b] = act[b] ? r : (T)0;
        });
        if constexpr (NW > 1) {
            double unused = 0.0;
            combine_waves(tot, unused);
        }
#pragma unroll
        for (int i = 0; i < NC; i++) acc[i] += tot[i];
    }
};

}  // namespace pmf
