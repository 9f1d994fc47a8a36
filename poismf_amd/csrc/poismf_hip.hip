// poismf_hip.hip -- the row kernels of the MI355X implementation of poismf's alternating factor-update path and their
// launchers.  Compiled once per inner solver and precision (-DPMF_TU=1 tncg / 2 cg / 3 pg / 4 the evaluation-only kernels of poismf_hip_debug_row_eval; without it all three, the
// -DPMF_TIMING development build) into libpoismf_hip_d.so / libpoismf_hip_f.so (-DUSE_FLOAT); the host side
// (sessions, run_poismf, planning) is poismf_hip_host.hip, the C-ABI is declared in include/poismf_hip.h.
//
// Device: one launch per (half-sweep, row bin); one wavefront per row, or 2 / 4 / 8 for longer rows
// (reg_eval.hpp, row_eval.hpp, solvers.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "plan.hpp"
#include "reg_eval.hpp"
#include "lane_eval.hpp"
#include "solvers.hpp"

// Translation units.  The row kernels of the three solvers are the bulk of the compile time, so the build compiles this
// file once per solver (poismf_amd/build.py): PMF_TU = K_TNCG / K_CG / K_PG instantiates the row kernels of one solver and
// exports one function, pmf_launch_one_tu<N>.  Without PMF_TU all three land in one translation unit.
#ifndef PMF_TU
#define PMF_TU -1
#endif
constexpr bool tu_has(int kmethod) { return PMF_TU == -1 || PMF_TU == kmethod; }

// Everything after the row's tile has been requested: starting point, per-row constant term, inner solver, store.
template <class EV, class T, int NC, int METHOD>
__device__ __forceinline__ void solve_row(const HalfArgs<T>& a, EV& ev, const T (&bs)[NC], unsigned lrow, unsigned nnz)
{
    const int k = a.geom.k;
    T* out = a.M + (size_t)(a.row_offset + lrow) * (size_t)k;
    T* out_p = a.Mp != nullptr ? a.Mp + (size_t)(a.row_offset + lrow) * (size_t)a.ldM : nullptr;
    T x[NC];
    if (nnz == 0) {  // rows without data are forced to zero every half (quirk Q7)
        // (built here, behind an opaque asm: as loop invariants of the row loop these zeros -- and wm1 below -- cost the
        // S = 28 instances their last registers and went to scratch)
        T z = (T)0;
        asm volatile("" : "+v"(z));
        PMF_EW x[i] = z;
        ev.store_vec(out, x);
        if (out_p != nullptr) ev.store_vec(out_p, x);
        return;
    }
    ev.start_point(out, x);
    ev.n_eval = 0;

    // per-row constant term: the k-vector itself, or (w != 1) the reference's Bsum_w row
    //   (w - 1) sum_j F_j + Bsum          ref: src/poismf.c:85-123 (adjustment_Bsum)
    T shift[NC];
    PMF_EW shift[i] = bs[i];
    const bool weighted = a.P.w != (T)1;
    if (weighted) {
        T cs[NC];
        PMF_EW cs[i] = (T)0;
        ev.tile_colsum(cs);
        T w_here = a.P.w;
        asm volatile("" : "+v"(w_here));
        const T wm1 = w_here - (T)1.;
        PMF_EW {
            shift[i] = cs[i] * wm1;
            shift[i] = shift[i] + bs[i];
        }
        if (METHOD == K_PG) {                                               // dscal_large, ref: :526, :576
            PMF_EW shift[i] = shift[i] * a.P.neg_step;
            // factors_multiple scales Bsum_w twice, one dscal_large after the other (ref: src/pred.c:121-122, :162)
            if (a.P.neg_step2 != (T)1) { PMF_EW shift[i] = shift[i] * a.P.neg_step2; }
        }
    }

    SolveStats st;
    if constexpr (METHOD == K_EVAL) {
        // G1 (plan.hpp, K_EVAL): the objective and gradient at the starting point, as the solvers' own wrappers compute them
        T g[NC];
        double f;
        if (a.P.maxupd == 0) {
            f = (double)fun_single(ev, a.P, shift, x);                  // ref: src/poismf.c:194-208
            grad_single(ev, a.P, shift, x, g, weighted);                // ref: :210-240
        } else f = (double)fun_and_grad(ev, a.P, shift, x, g);          // ref: :242-273 (no l2 term in f, quirk Q4)
        ev.store_vec(out, g);
        if (out_p != nullptr) ev.store_vec(out_p, g);
        if (a.dec_rows != nullptr && ev.lane == 0 && ev.wid == 0 && ev.member == 0) {
            const unsigned long long fb = __builtin_bit_cast(unsigned long long, f);
            a.dec_rows[2 * (size_t)lrow] = (unsigned)fb;
            a.dec_rows[2 * (size_t)lrow + 1] = (unsigned)(fb >> 32);
        }
        return;
    } else if constexpr (METHOD == K_PG) {
        pg_row(ev, a.P, x, shift);
    } else if constexpr (METHOD == K_CG) {
        // cached line search: streamed rows of the LDS engine (plan_geom decides), fp64 single-wave rows of the register engine
        if constexpr (EV::MAY_CACHE) {
            if (a.P.limit_step && ev.pq_cap > 0 && nnz <= (unsigned)ev.pq_cap) cg_row_cached(ev, a.P, shift, x, weighted, st);
            else cg_row(ev, a.P, shift, x, weighted, st);
        } else cg_row(ev, a.P, shift, x, weighted, st);
    } else {
        T prev[NC];
        PMF_EW prev[i] = x[i];
        if (!a.reuse_prev) { PMF_EW x[i] = (T)1e-3; }                   // ref: src/poismf.c:379-381
        (void)Tnc<T, NC, EV>::minimize(ev, a.P, shift, x, st);
        if (a.early_stop) {                                             // ref: src/poismf.c:393-396
            PMF_EW prev[i] = prev[i] - x[i];
            const T moved = ev.dot(prev, prev);
            if ((double)moved <= 1e-4 && ev.lane == 0 && ev.wid == 0 && ev.member == 0) atomicAdd(a.n_unchanged, 1u);
        }
    }
    ev.store_vec(out, x);
    if (out_p != nullptr) ev.store_vec(out_p, x);
    // per-row evaluation counts for the pass-weighted effective traffic (a plain read-modify-write: the row
    // has one owner; a shared counter here costs 4x the kernel time in contention)
#ifndef PMF_PROBE
    if (a.eval_rows != nullptr && ev.lane == 0 && ev.wid == 0 && ev.member == 0) a.eval_rows[lrow] += ev.n_eval;
#endif
    // the solver's decisions for this row (tests/test_gpu_decisions.py): { iterations | rc << 24, evaluations as the reference counts them }
    if (a.dec_rows != nullptr && ev.lane == 0 && ev.wid == 0 && ev.member == 0) {
        a.dec_rows[2 * (size_t)lrow] = (unsigned)st.niter | ((unsigned)st.rc << 24);
        a.dec_rows[2 * (size_t)lrow + 1] = (unsigned)st.nfeval;
    }
}

// One ticket from a launch's row queue -- or "past the end" once the call has been interrupted: HalfArgs::stop is a word in pinned host
// memory that the SIGINT handler itself sets, read (system scope) next to every ticket, as the reference's CG / TNCG row loops read
// should_stop_procedure before every row (ref: src/poismf.c:301, :360).  The read is issued before the atomic and travels beside it.
template <int METHOD, class T> __device__ __forceinline__ unsigned take_ticket(const HalfArgs<T>& a)
{
    unsigned st = 0u;
    if constexpr (METHOD != K_PG) {
        if (a.stop != nullptr) st = __hip_atomic_load(a.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const unsigned t = atomicAdd(a.queue, 1u);
    return st != 0u ? 0xffffffffu : t;
}

// A wavefront (or, NW > 1, a workgroup of NW wavefronts) walks rows blockIdx.x, blockIdx.x + gridDim.x, ... of the
// nnz-sorted permutation, or pulls them from a device-wide queue.
//
// Row hand-out.  PG does the same work for every nonzero, so the nnz-sorted rows are dealt out statically
// (row r, r + grid, ...).  CG and TNCG take anything from a handful to ~400 evaluations per row: there rows are
// pulled from a device-wide counter in nnz-descending order (longest first -- the GPU form of the reference's
// `schedule(dynamic)`, ref: src/poismf.c:296, :352), one returning atomic per row.
template <class EV, class T, int NC, int METHOD, int NW>
__device__ __forceinline__ void sweep_rows(const HalfArgs<T>& a, EV& ev, unsigned char* smem)
{
    // (the streamed re-run of a team launch: only if that launch gave up)
    if (a.gate != nullptr && __hip_atomic_load(a.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    ev.init(a.geom, a.F, smem);
    T bs[NC];
    ev.load_vec(a.bsum, bs);
    const RowDesc* desc = a.desc + a.perm_begin;
    if constexpr (EV::PIPELINED && (EV::PIPE_MW == 1 || (EV::PIPE_MW == 2 && METHOD == K_PG))) {
        // Software pipeline over the rows of this wave.  A row costs three dependent round trips to memory -- its
        // descriptor, its indices, the factor rows those name -- and the solver in between leaves the memory pipe idle.
        // Tickets (and descriptors) are fetched two rows ahead and the indices one row ahead, so that a row's gather
        // starts the moment the previous row is stored.
        unsigned r = blockIdx.x;
        auto ticket = [&]() -> unsigned {
            if (a.queue != nullptr) {
                if constexpr (NW > 1) {   // one atomic per workgroup, broadcast through LDS
                    unsigned* slot = ev.ticket_slot();
                    if (threadIdx.x == 0) *slot = take_ticket<METHOD>(a);
                    __syncthreads();
                    const unsigned t = uniform(*slot);
                    __syncthreads();
                    return t;
                }
                unsigned t = 0;
                if (ev.lane == 0) t = take_ticket<METHOD>(a);
                return uniform(t);
            }
            unsigned t = r;
            r += gridDim.x;
            if constexpr (METHOD != K_PG) {
                // (statically dealt rows: the stop word alone, one read of pinned host memory per row that nothing waits for until the ticket
                // is used, two rows later)
                if (a.stop != nullptr && __hip_atomic_load(a.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) t = 0xffffffffu;
            }
            return t;
        };
        auto fetch = [&](unsigned t) -> RowDesc { return desc[t < a.nrows ? t : 0u]; };
        unsigned t0 = ticket(), t1 = ticket();
        RowDesc d0 = fetch(t0), d1 = fetch(t1);
        if (t0 < a.nrows) ev.fetch_meta(a.indices + (((unsigned long long)d0.p0_hi << 32) | d0.p0_lo), d0.nnz);
#ifdef PMF_PROBE
        unsigned probe_i = 0;
#endif
        while (t0 < a.nrows) {
            const unsigned long long p0 = ((unsigned long long)d0.p0_hi << 32) | d0.p0_lo;
            if (d0.nnz != 0) ev.gather(a.values + p0, d0.nnz);                  // indices are here: request the tile
            const unsigned t2 = ticket();
            const RowDesc d2 = fetch(t2);
            if (t1 < a.nrows) ev.fetch_meta(a.indices + (((unsigned long long)d1.p0_hi << 32) | d1.p0_lo), d1.nnz);
#ifdef PMF_PROBE
            ev.probe = (a.eval_rows != nullptr && blockIdx.x == 5 && threadIdx.x < WAVE && probe_i < 60) ? a.eval_rows + 16 * probe_i : nullptr;
            if (ev.probe != nullptr && ev.lane == 0) { ev.probe[10] = (unsigned)__builtin_amdgcn_s_memtime(); ev.probe[11] = d0.nnz; }
            probe_i++;
#endif
            solve_row<EV, T, NC, METHOD>(a, ev, bs, d0.lrow, d0.nnz);
            t0 = t1; d0 = d1;
            t1 = t2; d1 = d2;
        }
        return;
    } else {
        unsigned r = blockIdx.x;
        for (;;) {
            if (a.queue != nullptr) {
                if constexpr (NW > 1) {
                    // broadcast through the last 16 bytes of the DYNAMIC LDS block (a static __shared__ object
                    // would shift the dynamic base off 16-byte alignment and slow every ds_read_b128 down)
                    unsigned* next_row = ev.ticket_slot();
                    if (threadIdx.x == 0) *next_row = take_ticket<METHOD>(a);
                    __syncthreads();
                    r = uniform(*next_row);
                    __syncthreads();
                } else {
                    unsigned t = 0;
                    if (ev.lane == 0) t = take_ticket<METHOD>(a);
                    r = uniform(t);
                }
            }
            if (r >= a.nrows) break;
            if constexpr (METHOD != K_PG) {
                if (a.queue == nullptr && a.stop != nullptr && uniform(__hip_atomic_load(a.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != 0u) break;
            }
            const RowDesc d = desc[r];
            r += gridDim.x;
            const unsigned nnz = uniform(d.nnz);
            const unsigned long long p0 = ((unsigned long long)uniform(d.p0_hi) << 32) | uniform(d.p0_lo);
            if (nnz != 0) ev.begin_row(a.indices + p0, a.values + p0, nnz);
            solve_row<EV, T, NC, METHOD>(a, ev, bs, uniform(d.lrow), nnz);
        }
    }
}

// LDS-tile engine (row_eval.hpp): one wavefront (= one 64-thread workgroup, so no workgroup barrier is ever needed and
// the LDS tile is private), or NW wavefronts per row for the long-row path.
template <class T, int NC, int METHOD, int SL, int NW>
__global__ __launch_bounds__(WAVE* NW) void half_sweep_kernel(const HalfArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // streamed rows prefetch the next chunk's tile (row_eval.hpp, PF); TNC's register budget is spent already
    // (only in the instances with a compile-time slot count: the generic fp64 CG instance is at 512 registers already,
    // and the 14 slots in flight pushed it into scratch -- and into wrong results on the k = 200 test)
    // (TNC: in the eight-wave long-row instances only -- a row of 1e5 nonzeros is ~800 chunks per evaluation, each a full trip to
    // memory when nothing is requested ahead; the one-wave instances have no registers left for the 14 slots in flight)
    constexpr bool PF = SL > 0 && METHOD != K_TNCG;
    RowEval<T, NC, SL, NW, PF> ev;
    if constexpr (NW > 1) {
        if (a.arrive != nullptr && threadIdx.x == 0) atomicAdd(a.arrive, 1u);
    }
#ifdef PMF_TIMING
    const unsigned long long t_kernel = __builtin_amdgcn_s_memtime();
#endif
    sweep_rows<RowEval<T, NC, SL, NW, PF>, T, NC, METHOD, NW>(a, ev, smem);
#ifdef PMF_TIMING
    ev.tacc[5] = __builtin_amdgcn_s_memtime() - t_kernel;
    if (ev.lane == 0 && ev.wid == 0)
        for (int q = 0; q < 6; q++) atomicAdd(&g_pmf_timing[q], ev.tacc[q]);
#endif
}

// Rows shared by a TEAM of workgroups whose members exchange their sums per evaluation (row_eval.hpp: team_sum).  Two engines take part: the
// streamed LDS engine for giant rows (RowEval TM: GT_M = 32 eight-wave workgroups per row above 8192 nonzeros) and the lane engine for k = 100
// fp64 rows of 385 .. 8192 nonzeros (LaneEval TM_: ceil(class / 384) four-wave workgroups, each keeping its share of the row RESIDENT).  A
// workgroup joins the next open team in ARRIVAL order (whatever the dispatcher and the other kernels on the chip do, a team's members are
// resident); the team's first member draws rows from the launch's queue (longest first) and posts each ticket in the team's mailbox, the others
// pick it up there.  Member m takes nonzeros [m S, (m + 1) S) of the row, S = ceil(nnz / M) rounded up to the engine's grain, as if they were a
// row of their own.  An exchange that times out sets the launch's error word: everybody leaves, and the host re-runs the launch's rows on the
// one-workgroup streamed kernel (as for the register teams).
template <class EV, class T, int NC, int METHOD>
__device__ __forceinline__ void team_rows(const HalfArgs<T>& a, EV& ev, unsigned char* smem)
{
    if (a.arrive != nullptr && threadIdx.x == 0) atomicAdd(a.arrive, 1u);
    ev.init(a.geom, a.F, smem);
    T bs[NC];
    ev.load_vec(a.bsum, bs);
    const RowDesc* desc = a.desc + a.perm_begin;
    const unsigned M = a.team_members;
    unsigned* box = ev.ticket_slot();
    if (threadIdx.x == 0) {
        const unsigned n = atomicAdd((unsigned*)a.team_buf, 1u);
        box[0] = n / M;
        box[1] = n % M;
    }
    __syncthreads();
    const unsigned team = uniform(box[0]);
    ev.member = (int)uniform(box[1]);
    __syncthreads();
    if ((size_t)(team + 1) * gt_team_words(M) + GT_HEAD_WORDS > GT_BUF_BYTES / 8) return;   // (the host never launches that many)
    ev.tm_M = (int)M;
    ev.tm_seq = 0;
    ev.tm_words = a.team_buf + GT_HEAD_WORDS + (size_t)team * gt_team_words(M);
    ev.tm_err = a.team_err;
    ev.tm_spin = a.team_spin;
    unsigned long long* mail = ev.tm_words;   // [2]: { row number << 32 | ticket } of the row with that parity; [2]: members that have arrived
    constexpr unsigned END = 0xffffffffu;
    // Nobody draws a row before the whole team is on the chip: behind the other bins' persistent workgroups (second stream, §4.8) a team's members can
    // arrive tens of milliseconds apart -- longer on bigger problems -- and an exchange waiting for a partner that has no CU yet would run into the
    // exchange's time-out.  The wait for arrivals is patient (64 x that limit) and ends at once when the launch has no rows left.
    if (threadIdx.x == 0) atomicAdd((unsigned*)(mail + 2), 1u);
    // (POISMF_HIP_TEAM_SPIN_LIMIT=1, the tests' way to a launch that gives up: it gives up HERE, always -- with a limit of one poll a fast leader's
    // ticket could still arrive in time, and a launch in which nobody happened to wait too long kept its team results: 5 of 12 runs under
    // POISMF_HIP_NO_ROW_INTERRUPT=1, whose tickets are a host-memory read faster)
    if (a.team_spin <= 1u && threadIdx.x == 0) __hip_atomic_store(a.team_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ev.wid == 0) {
        for (unsigned spins = 0;; spins++) {
            if (a.team_spin <= 1u) break;
            if (uniform((unsigned)gt_load(mail + 2)) >= M) break;
            if ((spins & 255u) == 255u) {
                if (uniform(__hip_atomic_load(a.team_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) break;
                if (uniform(__hip_atomic_load(a.queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= a.nrows) break;
            }
            if ((spins >> 6) > a.team_spin) { __hip_atomic_store(a.team_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    for (unsigned rowno = 1;; rowno++) {
        // (all 64 lanes of the first wave poll and post: no lane-divergent region around the waits -- row_eval.hpp, team_sum, on why)
        if (ev.wid == 0) {
            unsigned tk = END;
            if (ev.member == 0) {
                unsigned t0 = 0;
                if (ev.lane == 0) t0 = take_ticket<METHOD>(a);
                tk = uniform(t0);
                if (tk >= a.nrows || uniform(__hip_atomic_load(a.team_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) tk = END;
                gt_store(mail + (rowno & 1u), ((unsigned long long)rowno << 32) | tk);
            } else {
                // (a member whose leader has not arrived yet waits as long as other teams keep the queue moving or rows remain; the
                // time-out applies once nothing moves)
                unsigned seen = uniform(__hip_atomic_load(a.queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                for (unsigned spins = 0;; spins++) {
                    const unsigned long long v = gt_load(mail + (rowno & 1u));
                    const unsigned hi = uniform((unsigned)(v >> 32)), lo = uniform((unsigned)v);
                    if (hi == rowno) { tk = lo; break; }
                    if ((spins & 255u) == 255u) {
                        if (uniform(__hip_atomic_load(a.team_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) break;
                        const unsigned q = uniform(__hip_atomic_load(a.queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        if (q != seen) { seen = q; spins = 0; }
                        else if (rowno == 1u && q >= a.nrows) break;   // every row has an owner and this team never got a leader: nothing to do
                    }
                    if (spins > a.team_spin) { __hip_atomic_store(a.team_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            box[0] = tk;
        }
        __syncthreads();
        const unsigned t = uniform(box[0]);
        __syncthreads();
        if (t >= a.nrows) break;
        const RowDesc d = desc[t];
        const unsigned nnz = uniform(d.nnz);
        const unsigned long long p0 = ((unsigned long long)uniform(d.p0_hi) << 32) | uniform(d.p0_lo);
        const unsigned S = ((nnz + M - 1u) / M + (EV::TEAM_ROUND - 1u)) / EV::TEAM_ROUND * EV::TEAM_ROUND;
        const unsigned off = (unsigned)ev.member * S;
        const unsigned mine = off < nnz ? (nnz - off < S ? nnz - off : S) : 0u;
        ev.begin_row(a.indices + p0 + (mine ? off : 0u), a.values + p0 + (mine ? off : 0u), mine);
        solve_row<EV, T, NC, METHOD>(a, ev, bs, uniform(d.lrow), nnz);
    }
}
template <class T, int NC, int METHOD, int SL, int NW>
__global__ __launch_bounds__(WAVE* NW) void half_sweep_giant_kernel(const HalfArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool PF = SL > 0 && METHOD != K_TNCG;
    using EV = RowEval<T, NC, SL, NW, PF, true>;
    EV ev;
    team_rows<EV, T, NC, METHOD>(a, ev, smem);
}
template <class T, int METHOD, int KS, int LV, int LA, int LL, int NW, int LP>
__global__ __launch_bounds__(WAVE* NW) __attribute__((amdgpu_waves_per_eu(1, 1))) void half_sweep_lane_team_kernel(const HalfArgs<T> a)
{
    using EV = LaneEval<T, KS, LV, LA, LL, NW, false, LP, 0, true>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[EV::SMEM_BYTES];
    EV ev;
    team_rows<EV, T, EV::NC, METHOD>(a, ev, smem);
}

// Waves per SIMD the register allocator is asked to make room for: the largest count whose VGPR budget (512 per SIMD
// lane, granules of 8) holds the tile (TR = 4 S NS registers) plus an allowance for the solver.  The allowances are
// tuned on C2 (ms per sweep): PG 40 / 56 / 72 -> 1.32 / 1.15 / 1.31; CG 60 / 90 / 120 / 150 -> 3.68 / 3.38 / 3.31 / 3.53;
// TNCG 30 / 60 / 80 / 100 / 150 / 200 -> 57 / 26 / 17.5 / 16.6 / 18.4 / 20.2 (TNC keeps ~21 k-vectors: below its real
// need the idle ones spill around the evaluations, which is cheaper than giving up a wave -- up to a point).
#ifndef PMF_PG_EXTRA
#define PMF_PG_EXTRA 56
#endif
#ifndef PMF_CG_EXTRA
#define PMF_CG_EXTRA 120
#endif
#ifndef PMF_TNC_EXTRA
#define PMF_TNC_EXTRA 100
#endif
constexpr int reg_waves(int tile_regs, int method)
{
    const int need = tile_regs + (method == K_PG ? PMF_PG_EXTRA : method == K_CG ? PMF_CG_EXTRA : PMF_TNC_EXTRA);
    for (int w : { 8, 6, 5, 4, 3, 2 })
        if ((512 / w) / 8 * 8 >= need) return w;
    return 1;
}
// Register-tile engine (reg_eval.hpp) for rows of at most (64 / G) S nonzeros: no LDS at all, waves per CU set by VGPRs.
template <class T, int METHOD, int S, int G, int NS>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(reg_waves(4 * S * NS, METHOD)))) void half_sweep_reg_kernel(const HalfArgs<T> a)
{
    RegEval<T, S, G, NS> ev;
    constexpr int SMEM = RegEval<T, S, G, NS>::SMEM_BYTES;   // (only CG on doubles parks k-vectors here)
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM > 0 && METHOD == K_CG ? SMEM : 16];
#ifdef PMF_TIMING
    const unsigned long long t_kernel = __builtin_amdgcn_s_memtime();
#endif
    sweep_rows<RegEval<T, S, G, NS>, T, RegEval<T, S, G, NS>::NC, METHOD, 1>(a, ev, smem);
#ifdef PMF_TIMING
    if (ev.lane == 0) atomicAdd(&g_pmf_timing[5], __builtin_amdgcn_s_memtime() - t_kernel);
#endif
}

// The same engine with NW = 2, 4 or 8 wavefronts per row (reg_eval.hpp, NW_ > 1): rows of up to NW (64 / G) S nonzeros.
// The fewest waves whose shares fit their registers are used: every evaluation ends in a barrier and a round trip
// through LDS, which costs more the more waves take part (C2-shaped PG(10), ns per nonzero and half: 100-nonzero rows on
// one wave 0.06; 200-nonzero rows 0.137 on eight waves).
constexpr int regw_waves(int tile_regs, int method, int nw)
{
    // (PG: the cross-wave scratch pointers and wave index are 8 more registers than the one-wave kernel keeps; without the
    // allowance the S = 28 instances sat at 3 waves per SIMD with 20-28 bytes of scratch)
    const int w = reg_waves(tile_regs + (method == K_PG ? 8 : 0), method);
    return nw >= 16 && w < 4 ? 4 : (nw >= 8 && w < 2 ? 2 : w);   // eight waves are two per SIMD, sixteen four
}
template <class T, int METHOD, int S, int G, int NS, int NW>
__global__ __launch_bounds__(WAVE* NW) __attribute__((amdgpu_waves_per_eu(regw_waves(4 * S * NS, METHOD, NW)))) void half_sweep_regw_kernel(const HalfArgs<T> a)
{
    using EV = RegEval<T, S, G, NS, NW>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[EV::SMEM_BYTES];
    EV ev;
    sweep_rows<EV, T, EV::NC, METHOD, NW>(a, ev, smem);
}

// Teams of M workgroups per row (reg_eval.hpp, M_ > 1).  A workgroup joins the next open team of ITS XCD in arrival order
// (same XCD: the granules then travel through one L2), so whatever the dispatcher does, a team's members are on the chip.
// The team's first member draws the rows from the launch's queue and posts each ticket in the team's mailbox one row ahead;
// it starts a row only once the team is complete, and if the queue runs dry before that, it posts the end-of-rows ticket and
// leaves (late members find it and leave too).
template <class T, int METHOD, int S, int G, int NS, int NW, int M>
__global__ __launch_bounds__(WAVE* NW) __attribute__((amdgpu_waves_per_eu(1))) void half_sweep_team_kernel(const HalfArgs<T> a)
{
    using EV = RegEval<T, S, G, NS, NW, M>;
    constexpr int NC = EV::NC;
    __shared__ __attribute__((aligned(16))) unsigned char smem[EV::SMEM_BYTES];
    EV ev;
    ev.init(a.geom, a.F, smem);
    T bs[NC];
    ev.load_vec(a.bsum, bs);
    const RowDesc* desc = a.desc + a.perm_begin;
    unsigned* arrive = (unsigned*)a.team_buf;                 // [8] per XCD
    unsigned* err = a.team_err;
    unsigned* box = ev.ticket_slot();
    if (threadIdx.x == 0) {
        // a grid too small to leave M workgroups on every XCD forms its teams chip-wide (it is a multiple of M: no team
        // stays incomplete); a full-size grid may leave up to M - 1 workgroups per XCD without a team, idle until the rows run out
        unsigned xcc = 0;
        if (gridDim.x >= 8u * 4u * (unsigned)M) {
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            xcc &= 7u;
        }
        const unsigned n = atomicAdd(arrive + xcc, 1u);
        box[0] = xcc * TEAM_SLOTS_PER_XCD + n / (unsigned)M;
        box[1] = n % (unsigned)M;
        if (n / (unsigned)M >= TEAM_SLOTS_PER_XCD) __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned team = uniform(box[0]) % TEAM_SLOTS;
    ev.member = (int)uniform(box[1]);
    __syncthreads();
    ev.team_words = a.team_buf + TEAM_HEAD_WORDS + (size_t)team * TEAM_WORDS;
    ev.team_err = err;
    ev.team_spin = a.team_spin;
    unsigned long long* mail = ev.team_words;                 // [2]: { ticket, row number } of the row with that parity
    unsigned long long* here = ev.team_words + 2;             // [M_MAX]: member m has arrived
    constexpr unsigned END = 0xffffffffu;
    if (ev.member != 0 && threadIdx.x == 0) gran_store(here + ev.member, 1ull);

    // leader: wait for the team, or for the rows to run out.  A team that stays incomplete (a full-size grid leaves up to M - 1
    // workgroups per XCD without partners) is no error while other teams are draining the queue; if the queue does not move for
    // `team_spin` polls either, nothing is making progress: give the launch up (the host re-runs its rows on the streamed path).
    if (ev.member == 0) {
        if (threadIdx.x == 0) {
            unsigned full = 0;
            unsigned seen = __hip_atomic_load(a.queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (unsigned spins = 0; !full; spins++) {
                full = 1;
                for (int m = 1; m < M; m++) full &= gran_load(here + m) != 0ull;
                if (full) break;
                const unsigned q = __hip_atomic_load(a.queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (q >= a.nrows || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if (q != seen) { seen = q; spins = 0; }
                if (spins > a.team_spin) { __hip_atomic_store(err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(8);
            }
            box[0] = full;
        }
        __syncthreads();
        const bool full = uniform(box[0]) != 0;
        __syncthreads();
        if (!full) {
            if (threadIdx.x == 0) gran_store(mail + 1, ((unsigned long long)1u << 32) | END);   // the first row has number 1
            return;
        }
    }

    // rows: number 1, 2, ..; the ticket of row number i travels in mailbox slot i % 2.  Every member learns row i + 1's ticket
    // right after requesting row i's tile and fetches that row's indices while the solver runs (fetch_meta / gather, as the
    // single-workgroup kernels do in sweep_rows), so a row costs one trip to memory on the critical path, not three.
    auto next_ticket = [&](unsigned rowno_next) -> unsigned {     // (call with all threads)
        if (ev.member == 0) {
            if (threadIdx.x == 0) {
                box[0] = take_ticket<METHOD>(a);
                gran_store(mail + (rowno_next & 1u), ((unsigned long long)rowno_next << 32) | (box[0] < a.nrows ? box[0] : END));
            }
        } else if (threadIdx.x == 0) {
            // (a member of a team that is still incomplete waits for its FIRST ticket as long as other teams keep the queue moving:
            // its leader posts the end-of-rows ticket once the rows run out)
            unsigned long long v = 0;
            unsigned seen = __hip_atomic_load(a.queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (unsigned spins = 0;; spins++) {
                v = gran_load(mail + (rowno_next & 1u));
                if ((unsigned)(v >> 32) == rowno_next) break;
                if ((spins & 255u) == 255u) {
                    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { v = END; break; }
                    const unsigned q = __hip_atomic_load(a.queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (rowno_next == 1u && q != seen) { seen = q; spins = 0; }
                }
                if (spins > a.team_spin) {
                    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v = END;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            box[0] = (unsigned)v;
        }
        __syncthreads();
        const unsigned tk = uniform(box[0]);
        __syncthreads();
        return tk;
    };
    auto row_of = [&](unsigned tk, unsigned& nnz, unsigned long long& p0, unsigned& lrow) {
        const RowDesc d = desc[tk < a.nrows ? tk : 0u];
        nnz = uniform(d.nnz);
        p0 = ((unsigned long long)uniform(d.p0_hi) << 32) | uniform(d.p0_lo);
        lrow = uniform(d.lrow);
    };
    unsigned rowno = 1;
    unsigned t = next_ticket(1);
    unsigned nnz = 0, lrow = 0;
    unsigned long long p0 = 0;
    row_of(t, nnz, p0, lrow);
    if (t < a.nrows) ev.fetch_meta(a.indices + p0, nnz);
    while (t < a.nrows) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
#ifdef PMF_PROBE
        const unsigned long long t_row0 = __builtin_amdgcn_s_memtime();
        ev.probe_wait = 0;
#endif
        ev.gather(a.values + p0, nnz);                               // the indices are here: request the tile
        const unsigned t_next = next_ticket(rowno + 1);
        unsigned nnz_next = 0, lrow_next = 0;
        unsigned long long p0_next = 0;
        row_of(t_next, nnz_next, p0_next, lrow_next);
        if (t_next < a.nrows) ev.fetch_meta(a.indices + p0_next, nnz_next);
#ifdef PMF_PROBE
        {   // the tile has arrived when a value that depends on every load has
            T probe_sum[NC];
            PMF_EW probe_sum[i] = (T)0;
            ev.tile_touch(probe_sum);
            ev.probe_acc[4] += __builtin_amdgcn_s_memtime() - t_row0;
            if (probe_sum[0] == (T)123.456) ev.probe_acc[6]++;
        }
#endif
        solve_row<EV, T, NC, METHOD>(a, ev, bs, lrow, nnz);
#ifdef PMF_PROBE
        // (a probe build leaves the per-row counters to the stamps) low half: cycles / 256 spent waiting for the team, high half: of the whole row
        if (a.eval_rows != nullptr && threadIdx.x == 0 && ev.member == 0)
            a.eval_rows[lrow] = (unsigned)std::min<unsigned long long>(0xffffu, ev.probe_wait >> 8) |
                                ((unsigned)std::min<unsigned long long>(0xffffu, (__builtin_amdgcn_s_memtime() - t_row0) >> 8) << 16);
        ev.probe_acc[5] += __builtin_amdgcn_s_memtime() - t_row0;
        ev.probe_acc[6] += ev.probe_wait;
#endif
        rowno++;
        t = t_next; nnz = nnz_next; p0 = p0_next; lrow = lrow_next;
    }
#ifdef PMF_PROBE
    // kernel-wide sums of the first members' wave 0
    if (threadIdx.x == 0 && ev.member == 0)
        for (int q = 0; q < 8; q++) atomicAdd(a.team_buf + 8 + q, ev.probe_acc[q]);   // (free words of the head; last team launch wins)
#endif
}

// Lane-per-nonzero engine (lane_eval.hpp): doubles, 25 or 50 slots per factor row; NW waves per row; one wave per SIMD, or
// (SMALL: one register set, 14 KB of LDS per wave) two.
// (lane_two: instances compiled for two waves per SIMD -- the SMALL ones)
template <bool SMALL> constexpr bool lane_two() { return SMALL; }
template <class T, int METHOD, int KS, int LV, int LA, int LL, int NW, bool SMALL, int LP = 0, int TX = 0>
__global__ __launch_bounds__(WAVE* NW) __attribute__((amdgpu_waves_per_eu(lane_two<SMALL>() ? 2 : 1, lane_two<SMALL>() ? 2 : 1))) void half_sweep_lane_kernel(const HalfArgs<T> a)
{
    using EV = LaneEval<T, KS, LV, LA, LL, NW, SMALL, LP, TX>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[EV::SMEM_BYTES];
    EV ev;
    sweep_rows<EV, T, EV::NC, METHOD, NW>(a, ev, smem);
}

namespace {

constexpr int MAX_DEVICES = 64;        // per-device caches of kernel attributes below
// device and CU count of the launch being issued by this thread (set by launch_one_here from the session's values)
thread_local int t_device = 0, t_num_cu = 256;

template <int NC, int METHOD, int SL, int NW> int launch_bin(hipStream_t stream, const HalfArgs<real_t>& a, size_t lds, unsigned grid)
{
    auto kern = half_sweep_kernel<real_t, NC, METHOD, SL, NW>;
    // hipFuncAttributeMaxDynamicSharedMemorySize is per device: remember it per device id
    static std::atomic<bool> attr_set[MAX_DEVICES];
    const int dev = t_device >= 0 && t_device < MAX_DEVICES ? t_device : 0;
    if (!attr_set[dev].load(std::memory_order_acquire) || dev != t_device) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)LDS_PER_CU));
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE * NW), lds, stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int NC, int METHOD, int SL> int launch_giant(hipStream_t stream, const HalfArgs<real_t>& a, size_t lds, unsigned grid)
{
    auto kern = half_sweep_giant_kernel<real_t, NC, METHOD, SL, LONG_NW>;
    static std::atomic<bool> attr_set[MAX_DEVICES];
    const int dev = t_device >= 0 && t_device < MAX_DEVICES ? t_device : 0;
    if (!attr_set[dev].load(std::memory_order_acquire) || dev != t_device) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_PER_CU));
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE * LONG_NW), lds, stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}
// giant-row teams: TNCG only (the solver whose evaluations re-stream the row; CG has its cached line search and the register teams)
template <int NC, int SL> int launch_giant_method(hipStream_t stream, int method, const HalfArgs<real_t>& a, size_t lds, unsigned grid)
{
    if constexpr (tu_has(K_TNCG)) {
        if (method == POISMF_TNCG) return launch_giant<NC, K_TNCG, SL>(stream, a, lds, grid);
    }
    return 1;
}

template <int NC, int SL, int NW = 1> int launch_method(hipStream_t stream, int method, const HalfArgs<real_t>& a, size_t lds, unsigned grid)
{
    switch (method) {
        case POISMF_PG:
            if constexpr (tu_has(K_PG)) return launch_bin<NC, K_PG, SL, NW>(stream, a, lds, grid);
            else return 1;
        case POISMF_CG:
            if constexpr (tu_has(K_CG)) return launch_bin<NC, K_CG, SL, NW>(stream, a, lds, grid);
            else return 1;
        case POISMF_EVAL:
            if constexpr (tu_has(K_EVAL)) return launch_bin<NC, K_EVAL, SL, NW>(stream, a, lds, grid);
            else return 1;
        default:
            if constexpr (tu_has(K_TNCG)) return launch_bin<NC, K_TNCG, SL, NW>(stream, a, lds, grid);
            else return 1;
    }
}

template <int METHOD, int S, int NS> int launch_reg(hipStream_t stream, const HalfArgs<real_t>& a, unsigned grid_mult)
{
    auto kern = half_sweep_reg_kernel<real_t, METHOD, S, REG_G, NS>;
    static std::atomic<int> occ_cache{0};  // waves per CU the register budget of this instance allows (a property of the code object)
    int occ = occ_cache.load(std::memory_order_relaxed);
    if (occ == 0) {
        int n = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kern), WAVE, 0));
        occ = std::max(1, n);
        occ_cache.store(occ, std::memory_order_relaxed);
    }
    const unsigned grid = (unsigned)std::min<size_t>(a.nrows, (size_t)t_num_cu * (size_t)occ * grid_mult);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int S, int NS> int launch_reg_method(hipStream_t stream, int method, const HalfArgs<real_t>& a, unsigned grid_mult)
{
    switch (method) {
        case POISMF_PG:
            if constexpr (tu_has(K_PG)) return launch_reg<K_PG, S, NS>(stream, a, grid_mult);
            else return 1;
        case POISMF_CG:
            if constexpr (tu_has(K_CG) && S * REG_JG <= REG_NNZ_MAX_CG && (NS == 1 || S * REG_JG <= 144)) return launch_reg<K_CG, S, NS>(stream, a, grid_mult);
            else return 1;
        case POISMF_EVAL:   // (the sizes CG has)
            if constexpr (tu_has(K_EVAL) && S * REG_JG <= REG_NNZ_MAX_CG && (NS == 1 || S * REG_JG <= 144)) return launch_reg<K_EVAL, S, NS>(stream, a, grid_mult);
            else return 1;
        default:
            if constexpr (tu_has(K_TNCG) && S * REG_JG <= REG_NNZ_MAX_TNCG && (NS == 1 || S * REG_JG <= 112)) return launch_reg<K_TNCG, S, NS>(stream, a, grid_mult);
            else return 1;
    }
}

template <int METHOD, int S, int NS, int NW> int launch_regw(hipStream_t stream, const HalfArgs<real_t>& a, unsigned grid_mult)
{
    auto kern = half_sweep_regw_kernel<real_t, METHOD, S, REG_G, NS, NW>;
    static std::atomic<int> occ_cache{0};  // workgroups per CU
    int occ = occ_cache.load(std::memory_order_relaxed);
    if (occ == 0) {
        int n = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kern), WAVE * NW, 0));
        occ = std::max(1, n);
        occ_cache.store(occ, std::memory_order_relaxed);
    }
    const unsigned grid = (unsigned)std::min<size_t>(a.nrows, (size_t)t_num_cu * (size_t)occ * grid_mult);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE * NW), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// team launches: CG on doubles with two slots per lane (k = 50 fp64) is what asks for them
template <int M, int S> int launch_team(hipStream_t stream, int method, const HalfArgs<real_t>& a)
{
    if constexpr ((tu_has(K_CG) || tu_has(K_EVAL)) && sizeof(real_t) == 8 && REG_G == 16) {
        constexpr int KM = tu_has(K_CG) ? K_CG : K_EVAL;   // (the translation unit of the evaluation-only kernels has no CG)
        if (method != (KM == K_CG ? POISMF_CG : POISMF_EVAL)) return 1;
        auto kern = half_sweep_team_kernel<real_t, KM, S, REG_G, 2, TEAM_NW, M>;
        // as many workgroups as CUs: each takes a CU's whole register file (one wave of 512 per SIMD)
        const unsigned grid = (unsigned)std::min<size_t>((size_t)a.nrows * M, (size_t)t_num_cu / M * M);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE * TEAM_NW), 0, stream, a);
        HIP_TRY(hipGetLastError());
        return 0;
    } else return 1;
}

// lane-per-nonzero launches
template <int METHOD, int KS, int LV, int LA, int LL, int NW, bool SMALL = false, int LP = 0, int TX = 0> int launch_lane(hipStream_t stream, const HalfArgs<real_t>& a, unsigned grid_mult)
{
    if constexpr (tu_has(METHOD)) {
        using EV = LaneEval<real_t, KS, LV, LA, LL, NW, SMALL, LP, TX>;
        auto kern = half_sweep_lane_kernel<real_t, METHOD, KS, LV, LA, LL, NW, SMALL, LP, TX>;
        // workgroups per CU: one (SMALL: two) waves per SIMD, and the LDS each takes
        const int occ = std::max(1, std::min((lane_two<SMALL>() ? 8 : 4) / NW, (int)(LDS_PER_CU / (size_t)EV::SMEM_BYTES)));
        const unsigned grid = (unsigned)std::min<size_t>(a.nrows, (size_t)t_num_cu * (size_t)occ * grid_mult);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE * NW), 0, stream, a);
        HIP_TRY(hipGetLastError());
        return 0;
    } else return 1;
}
// lane teams: k = 100 fp64 TNCG rows of 385 .. 8192 nonzeros, M = a.team_members four-wave workgroups per row
int launch_lane_team(hipStream_t stream, int method, int s_load, int lv, int la, int ll, int nw, int lp, const HalfArgs<real_t>& a, unsigned grid)
{
    if constexpr (tu_has(K_TNCG) && sizeof(real_t) == 8) {
        if (method == POISMF_TNCG && s_load == 50 && lv == 1 && la == 0 && ll == 0 && nw == 4 && lp == 32) {
            hipLaunchKernelGGL((half_sweep_lane_team_kernel<real_t, K_TNCG, 50, 1, 0, 0, 4, 32>), dim3(grid), dim3(WAVE * 4), 0, stream, a);
            HIP_TRY(hipGetLastError());
            return 0;
        }
    }
    return 1;
}
template <int METHOD> int launch_lane_shape(hipStream_t stream, int s_load, int lv, int la, int ll, int nw, int small, int lp, int tx, const HalfArgs<real_t>& a, unsigned grid_mult)
{
    const int key = (((lv * 10 + la) * 10 + ll) * 10 + nw) * 10 + small + (lp > 0 ? 100000 : 0);
    if constexpr (sizeof(real_t) == 8) {
        if constexpr (METHOD == K_PG) return 1;
        else if (s_load == 25) {
            switch (key) {
                case 10011: return launch_lane<METHOD, 25, 1, 0, 0, 1, true>(stream, a, grid_mult);
                case 10110: return launch_lane<METHOD, 25, 1, 0, 1, 1>(stream, a, grid_mult);
                case 12110: return launch_lane<METHOD, 25, 1, 2, 1, 1>(stream, a, grid_mult);
                case 12120: return launch_lane<METHOD, 25, 1, 2, 1, 2>(stream, a, grid_mult);
                case 12140: return launch_lane<METHOD, 25, 1, 2, 1, 4>(stream, a, grid_mult);
                case 112140: return launch_lane<METHOD, 25, 1, 2, 1, 4, false, 16>(stream, a, grid_mult);
                case 110011:
                    if constexpr (METHOD != K_TNCG) {
                        if (lp == 32) return launch_lane<METHOD, 25, 1, 0, 0, 1, true, 32>(stream, a, grid_mult);
                    }
                    break;
            }
        } else if (s_load == 50) {
            if (tx == 48 && key == 10010) return launch_lane<METHOD, 50, 1, 0, 0, 1, false, 0, 48>(stream, a, grid_mult);
            if (tx == 64 && key == 10010) return launch_lane<METHOD, 50, 1, 0, 0, 1, false, 0, 64>(stream, a, grid_mult);
            switch (key) {
                case 10020: return launch_lane<METHOD, 50, 1, 0, 0, 2>(stream, a, grid_mult);
                case 110040: if (lp == 32) return launch_lane<METHOD, 50, 1, 0, 0, 4, false, 32>(stream, a, grid_mult); break;
            }
        }
    } else {
        if (s_load == 13) {
            if constexpr (METHOD == K_PG) {
                switch (key) {
                    case 40041: return launch_lane<METHOD, 13, 4, 0, 0, 4, true>(stream, a, grid_mult);
                    case 140041: if (lp == 16) return launch_lane<METHOD, 13, 4, 0, 0, 4, true, 16>(stream, a, grid_mult); break;
                }
            } else {
                switch (key) {
                    case 10011: return launch_lane<METHOD, 13, 1, 0, 0, 1, true>(stream, a, grid_mult);
                    case 20011: return launch_lane<METHOD, 13, 2, 0, 0, 1, true>(stream, a, grid_mult);
                    case 20021: return launch_lane<METHOD, 13, 2, 0, 0, 2, true>(stream, a, grid_mult);
                    case 20041: return launch_lane<METHOD, 13, 2, 0, 0, 4, true>(stream, a, grid_mult);
                    case 20081: return launch_lane<METHOD, 13, 2, 0, 0, 8, true>(stream, a, grid_mult);
                    case 30081: return launch_lane<METHOD, 13, 3, 0, 0, 8, true>(stream, a, grid_mult);
                }
            }
        }
    }
    return 1;
}

template <int S, int NS, int NW> int launch_regw_method(hipStream_t stream, int method, const HalfArgs<real_t>& a, unsigned grid_mult)
{
    if constexpr (S * REG_JG < 32) return 1;
    else if constexpr (NW == 16 && (S * REG_JG > REGW16_WAVE_NNZ || NS != 1)) return 1;
    else switch (method) {
        case POISMF_PG:
            if constexpr (tu_has(K_PG)) return launch_regw<K_PG, S, NS, NW>(stream, a, grid_mult);
            else return 1;
        case POISMF_CG:
            if constexpr (tu_has(K_CG) && S * REG_JG <= REGW_WAVE_NNZ_MAX_CG && NW <= 8) return launch_regw<K_CG, S, NS, NW>(stream, a, grid_mult);
            else return 1;
        case POISMF_EVAL:
            if constexpr (tu_has(K_EVAL) && S * REG_JG <= REGW_WAVE_NNZ_MAX_CG && NW <= 8) return launch_regw<K_EVAL, S, NS, NW>(stream, a, grid_mult);
            else return 1;
        default:
            if constexpr (tu_has(K_TNCG) && S * REG_JG <= REGW_WAVE_NNZ_MAX_TNCG && NW <= 8) return launch_regw<K_TNCG, S, NS, NW>(stream, a, grid_mult);
            else return 1;
    }
}

template <int NS, int NW> int launch_regw_steps_nw(hipStream_t stream, int S, int method, const HalfArgs<real_t>& a, unsigned grid_mult)
{
    switch (S) {
#define X(NZ) case (NZ) / REG_JG: return launch_regw_method<(NZ) / REG_JG, NS, NW>(stream, method, a, grid_mult);
        PMF_REG_SIZES(X)
#undef X
    }
    return 1;
}
template <int NS> int launch_regw_steps(hipStream_t stream, int nw, int S, int method, const HalfArgs<real_t>& a, unsigned grid_mult)
{
    switch (nw) {
        case 2: return launch_regw_steps_nw<NS, 2>(stream, S, method, a, grid_mult);
        case 4: return launch_regw_steps_nw<NS, 4>(stream, S, method, a, grid_mult);
        case 8: return launch_regw_steps_nw<NS, 8>(stream, S, method, a, grid_mult);
        case 16:
            if constexpr (PMF_REGW16) return launch_regw_steps_nw<NS, 16>(stream, S, method, a, grid_mult);
            else return 1;
    }
    return 1;
}

template <int NS> int launch_reg_steps(hipStream_t stream, int S, int method, const HalfArgs<real_t>& a, unsigned grid_mult)
{
    switch (S) {
#define X(NZ) case (NZ) / REG_JG: return launch_reg_method<(NZ) / REG_JG, NS>(stream, method, a, grid_mult);
        PMF_REG_SIZES(X)
#undef X
    }
    return 1;
}


int launch_one_here(int method, const OneLaunch& o, const HalfArgs<real_t>& a)
{
    int rc = 1;
    t_device = o.device; t_num_cu = o.num_cu;
    if (o.lane_L > 0) {
        if (o.team > 1) return launch_lane_team(o.main_stream, method, o.s_load, o.lane_L, o.lane_A, o.lane_LL, o.nw, o.lane_LP, a, o.grid);
        if (method == POISMF_PG) return launch_lane_shape<K_PG>(o.main_stream, o.s_load, o.lane_L, o.lane_A, o.lane_LL, o.nw, o.lane_small, o.lane_LP, o.lane_tx, a, o.grid_mult);
        if (method == POISMF_CG) return launch_lane_shape<K_CG>(o.nw > 1 ? o.main_stream : o.bin_stream, o.s_load, o.lane_L, o.lane_A, o.lane_LL, o.nw, o.lane_small, o.lane_LP, o.lane_tx, a, o.grid_mult);
        if (method == POISMF_TNCG) return launch_lane_shape<K_TNCG>(o.nw > 1 ? o.main_stream : o.bin_stream, o.s_load, o.lane_L, o.lane_A, o.lane_LL, o.nw, o.lane_small, o.lane_LP, o.lane_tx, a, o.grid_mult);
        if (method == POISMF_EVAL) return launch_lane_shape<K_EVAL>(o.nw > 1 ? o.main_stream : o.bin_stream, o.s_load, o.lane_L, o.lane_A, o.lane_LL, o.nw, o.lane_small, o.lane_LP, o.lane_tx, a, o.grid_mult);
        return 1;
    }
#ifdef PMF_LANE_ONLY   // development: compile the lane-per-nonzero kernels alone (seconds instead of minutes)
    return rc;
#else
    if (o.team > 1 && o.team != GT_M) {
        if (o.team == 2 && o.reg_S == 32) return launch_team<2, 32>(o.main_stream, method, a);
        if (o.team == 2 && o.reg_S == 36) return launch_team<2, 36>(o.main_stream, method, a);
        if (o.team == 3 && o.reg_S == 28) return launch_team<3, 28>(o.main_stream, method, a);
        if (o.team == 3 && o.reg_S == 32) return launch_team<3, 32>(o.main_stream, method, a);
        if (o.team == 4 && o.reg_S == 32) return launch_team<4, 32>(o.main_stream, method, a);
        return 1;
    }
    if (o.reg_S > 0) {
        if (o.nw > 1) {
            if constexpr (REG_G == 16) rc = launch_regw_steps<1>(o.main_stream, o.nw, o.reg_S, method, a, o.grid_mult);
            else rc = o.s_load <= REG_G ? launch_regw_steps<1>(o.main_stream, o.nw, o.reg_S, method, a, o.grid_mult)
                                        : launch_regw_steps<2>(o.main_stream, o.nw, o.reg_S, method, a, o.grid_mult);
        } else if constexpr (REG_NS_MAX == 1) rc = launch_reg_steps<1>(o.bin_stream, o.reg_S, method, a, o.grid_mult);
        else rc = o.s_load <= REG_G ? launch_reg_steps<1>(o.bin_stream, o.reg_S, method, a, o.grid_mult)
                                    : launch_reg_steps<2>(o.bin_stream, o.reg_S, method, a, o.grid_mult);
        return rc;
    }
    if (o.nw > 1 && o.team == GT_M) {   // giant-row teams (row_eval.hpp, TM)
        if (!o.generic_only && o.s_load == SPECIAL_SL_A) return launch_giant_method<SLOT_ELEMS, SPECIAL_SL_A>(o.long_stream, method, a, o.lds, o.grid);
        if (!o.generic_only && o.s_load == SPECIAL_SL_B) return launch_giant_method<SLOT_ELEMS, SPECIAL_SL_B>(o.long_stream, method, a, o.lds, o.grid);
        switch (o.spl) {
            case 1: return launch_giant_method<1 * SLOT_ELEMS, 0>(o.long_stream, method, a, o.lds, o.grid);
            case 2: return launch_giant_method<2 * SLOT_ELEMS, 0>(o.long_stream, method, a, o.lds, o.grid);
        }
        return 1;
    }
    if (o.nw > 1) {
        // (the long-row path takes the compile-time slot counts of the BASELINE configs too, and with them the prefetch of the next chunk)
        if (!o.generic_only && o.s_load == SPECIAL_SL_A) rc = launch_method<SLOT_ELEMS, SPECIAL_SL_A, LONG_NW>(o.long_stream, method, a, o.lds, o.grid);
        else if (!o.generic_only && o.s_load == SPECIAL_SL_B) rc = launch_method<SLOT_ELEMS, SPECIAL_SL_B, LONG_NW>(o.long_stream, method, a, o.lds, o.grid);
        else switch (o.spl) {
            case 1: rc = launch_method<1 * SLOT_ELEMS, 0, LONG_NW>(o.long_stream, method, a, o.lds, o.grid); break;
            case 2: rc = launch_method<2 * SLOT_ELEMS, 0, LONG_NW>(o.long_stream, method, a, o.lds, o.grid); break;
        }
        return rc;
    }
    if (!o.generic_only && o.s_load == SPECIAL_SL_A) rc = launch_method<SLOT_ELEMS, SPECIAL_SL_A>(o.bin_stream, method, a, o.lds, o.grid);
    else if (!o.generic_only && o.s_load == SPECIAL_SL_B) rc = launch_method<SLOT_ELEMS, SPECIAL_SL_B>(o.bin_stream, method, a, o.lds, o.grid);
    else switch (o.spl) {
        case 1: rc = launch_method<1 * SLOT_ELEMS, 0>(o.bin_stream, method, a, o.lds, o.grid); break;
        case 2: rc = launch_method<2 * SLOT_ELEMS, 0>(o.bin_stream, method, a, o.lds, o.grid); break;
    }
    return rc;
#endif
}
}  // namespace

#if PMF_TU == 1 || PMF_TU == -1
int pmf_launch_one_tu1(int method, const OneLaunch& o, const HalfArgs<real_t>& a) { return launch_one_here(method, o, a); }
#endif
#if PMF_TU == 2 || PMF_TU == -1
int pmf_launch_one_tu2(int method, const OneLaunch& o, const HalfArgs<real_t>& a) { return launch_one_here(method, o, a); }
#endif
#if PMF_TU == 3 || PMF_TU == -1
int pmf_launch_one_tu3(int method, const OneLaunch& o, const HalfArgs<real_t>& a) { return launch_one_here(method, o, a); }
#endif
#if PMF_TU == 4 || PMF_TU == -1
int pmf_launch_one_tu4(int method, const OneLaunch& o, const HalfArgs<real_t>& a) { return launch_one_here(method, o, a); }
#endif

#ifdef PMF_TIMING
// development-only (the phase timers live in this translation unit): read and reset them
int pmf_read_timing(unsigned long long* out)
{
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pmf_timing), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pmf_timing), z, sizeof(z)) != hipSuccess;
}
#endif
