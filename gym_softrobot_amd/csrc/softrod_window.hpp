// softrod_window.hpp — rods of 64..102 elements on TWO wavefronts with overlapping windows
// (BASELINE.json configs[2]: OctoArmSingle at 100 elements).
//
// The one-wave form of such a rod keeps two nodes per lane and needs the whole 512-entry
// register file: one wave per SIMD, 10 % of its instructions are accumulator-register copies
// (profiles/README.md).  Here the rod is stepped by a workgroup of two waves at ONE node per
// lane — the code and the register budget of the 50-element rod — without a per-stencil halo
// exchange: wave 0 holds nodes 0..63, wave 1 holds nodes n-63..n, so the two windows overlap
// by 127-n nodes and 128 lanes carry n+1 nodes with the surplus as halo instead of idle.
// Each wave owns its half of the rod and recomputes a copy of the other's edge.  The outermost
// lane of a window has no neighbour, so its value is wrong after one substep, and the error
// creeps inward by RHO nodes per substep (the reach of one substep's stencils, contact
// included).  Before it can touch an owned node the halo is overwritten with the owner's
// values through LDS — every `refresh` substeps, two barriers — which makes the scheme EXACT:
// every owned node sees the same operands in the same order as on a single long wave.
//
// The kernel does the prologue (set_action) and the substeps; reward / observation come from
// the ordinary two-slot kernel launched with n_sub = 0 on the same rows (the layout is the
// same: slot = node index, 128 slots per rod).
#pragma once

namespace softrod {

// Nodes of contamination per substep, rounded up: with a 13-node halo and unfrozen edges the
// owned nodes stay bit-clean for a refresh interval of 4 substeps and not for 6, i.e. the
// front moves between 2.2 and 3.25 nodes per substep (contact's two averaging rounds included).
constexpr int kWindowRho = 4;

template <unsigned F>
__device__ __forceinline__ void window_load(const StatePtrs& S, size_t N, int rod, int g, LaneN<1>& L) {
    constexpr size_t W = 2 * kLanes;
    const size_t m = (size_t)rod * W + (size_t)g;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        L.x[0][c] = S.pos[c * N * W + m];
        L.v[0][c] = S.vel[c * N * W + m];
        L.w[0][c] = S.omg[c * N * W + m];
        L.t[0][c] = S.tan[c * N * W + m];
        L.kap[0][c] = S.kap[c * N * W + m];
        L.rk[0][c] = S.rkap[c * N * W + m];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) L.Q[0][c] = S.dir[c * N * W + m];
}

__device__ __forceinline__ void window_store(const StatePtrs& S, size_t N, int rod, int g, const LaneN<1>& L) {
    constexpr size_t W = 2 * kLanes;
    const size_t m = (size_t)rod * W + (size_t)g;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        S.pos[c * N * W + m] = L.x[0][c];
        S.vel[c * N * W + m] = L.v[0][c];
        S.omg[c * N * W + m] = L.w[0][c];
        S.tan[c * N * W + m] = L.t[0][c];
        S.kap[c * N * W + m] = L.kap[0][c];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) S.dir[c * N * W + m] = L.Q[0][c];
}

// grid = ceil(n_envs / RPB), block = 128 RPB.  ArmSingle feature set (no boundary condition, no filter).
// RPB = rods per workgroup.  RPB = 1: the rod's two waves meet at s_barrier (twice per refresh).
// RPB = 4: FOUR rods per workgroup of eight waves, rod e on waves e and e + 4 — which gfx950 places
// on the SAME SIMD (probed with s_getreg HW_ID, softrod_octo.hpp) — so that while one window waits
// for the other's halo the other is what the SIMD runs, and the rendezvous is a counter in LDS
// (release / acquire at workgroup scope) on double-buffered halo rows instead of a workgroup
// barrier that would couple four unrelated rods: exchange k writes buffer k & 1 and raises the
// wave's counter to k + 1; the partner cannot be more than one exchange ahead (it waits for this
// wave's counter before it reads), so buffer k & 1 is free again by the time exchange k + 2 writes it.
#ifndef SOFTROD_WINDOW_PRIO
#define SOFTROD_WINDOW_PRIO 1
#endif
template <unsigned F, int RPB = 1>
__global__ void __launch_bounds__(2 * kLanes * RPB, SOFTROD_CONTACT_WAVES)
softrod_step_window_kernel(const RodParams P, const StatePtrs S, const float* __restrict__ actions,
                           const int n_sub, const int refresh) {
    static_assert(RPB == 1 || RPB == 4, "one rod per workgroup, or four with partner waves on one SIMD");
    constexpr int kFields = 18;
    constexpr int kHalo = (RPB == 1) ? kLanes : 32;          // lanes a window can send (<= 31 for n >= 64)
    constexpr int kBuf = (RPB == 1) ? 1 : 2;
    __shared__ double ex_[RPB][kBuf][2][kFields][kHalo];
    __shared__ int flag_[RPB][2];
    __shared__ int sany_[RPB];

    const int wib = threadIdx.x >> 6;                                  // wave in the block
    const int es = (RPB == 1) ? 0 : (wib & (RPB - 1));                 // rod slot in the block
    const int wave = (RPB == 1) ? wib : (wib / RPB);                   // window of the rod
    const int lane = threadIdx.x & 63;
    const int tid = wave * kLanes + lane;
    const bool active = (RPB == 1) || ((int)blockIdx.x * RPB + es < P.n_envs);
    const int rod = active ? (int)blockIdx.x * RPB + es : P.n_envs - 1;   // idle slots shadow a real rod, read-only
    auto& ex = ex_[es];
    const size_t N = (size_t)P.n_envs;
    const int n = P.n_elem;
    bool live = active && !(S.skip && S.skip[rod]);   // (skip: reset by the auto-reset pass; the epilogue launch clears the flag)
    if constexpr (RPB == 1) { if (!live) return; }
    const int off = wave ? (n + 1 - kLanes) : 0;   // first node of this wave's window
    const int g = off + lane;                      // this lane's node / element / Voronoi index
    const int split = (n + 1) / 2;                 // wave 0 owns nodes < split, wave 1 the rest
    const bool owned = wave ? (g >= split) : (g < split);
    // The last lane of wave 0 is an interior node whose "next" shifts in zeros.  It is given the
    // index of the rod's end node, so that its element / Voronoi vertex are inert (no stiffness,
    // excluded from the wave-uniform range checks) instead of producing an absurd strain.
    const int gi = (wave == 0 && lane == kLanes - 1) ? n : g;
    if constexpr (RPB > 1) {
        if (tid < 2) flag_[es][tid] = 0;
        if (tid == 0) sany_[es] = 0;
    }

    LaneN<1> L;
    window_load<F>(S, N, rod, g, L);
    if (actions) {    // set_action (arm_single_env.py:226-235): rest_kappa[0,:] = basis @ action
        double rk0 = 0.0;
        if (g < n - 1) {
            const double* wrow = S.basis + (size_t)g * 7;
#pragma unroll
            for (int j = 0; j < 7; ++j) rk0 += wrow[j] * (double)actions[7 * (size_t)rod + j];
        }
        L.rk[0][0] = rk0;
    }
    EnvAction A;
#pragma unroll
    for (int i = 0; i < 7; ++i) A.a[i] = 0.0f;
    A.force = 0.0;
    ConstN<1> C;
    build_const<F, 1>(P, gi, A, C);
    // The two lanes at the inner ends of the windows have no neighbour on one side.  Integrating
    // them with the resulting unbalanced loads would, within a few substeps, throw their rates
    // far enough off to drag the whole wave into the range-reduction slow paths.  They are
    // frozen instead: their state stays what the last refresh delivered (stale by at most
    // `refresh` substeps), and the staleness — not an imbalance — is what creeps inward.
    if ((wave == 0 && lane == kLanes - 1) || (wave == 1 && lane == 0)) {
        C.hx[0] = 0.0; C.hq[0] = 0.0; C.cf[0] = 0.0; C.cw01[0] = 0.0; C.cw2[0] = 0.0;
        C.ca[0][0] = 0.0; C.ca[0][1] = 0.0; C.ca[0][2] = 0.0;
    }
    BcTargets B;
#pragma unroll
    for (int i = 0; i < 3; ++i) { B.pos[i] = 0.0; B.vel[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < 9; ++i) B.Q[i] = 0.0;
    RodParams Pk = P;
    if (!has<F>(P, SOFTROD_FEAT_ANALYTICAL_DAMPER)) Pk.damp_t = 1.0;
    double time = S.time[rod];

    // halo: wave 0 receives nodes split..63 from wave 1, wave 1 receives off..split-1 from wave 0
    const bool send = wave ? (g >= split && g < kLanes) : (g >= (n + 1 - kLanes) && g < split);
    const bool recv = !owned;
    int round = 0;            // RPB > 1: exchanges done so far
    auto exchange = [&]() {
        const int b = (RPB == 1) ? 0 : (round & 1);
        if (send) {
            const int j = wave ? g - split : g - (n + 1 - kLanes);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                ex[b][wave][c][j] = L.x[0][c];
                ex[b][wave][3 + c][j] = L.v[0][c];
                ex[b][wave][6 + c][j] = L.w[0][c];
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) ex[b][wave][9 + c][j] = L.Q[0][c];
        }
        if constexpr (RPB == 1) __syncthreads();
        else {
            // this wave's LDS writes are in order: the counter goes up after the halo rows are there
            if (lane == 0)
                __hip_atomic_store(&flag_[es][wave], round + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(&flag_[es][wave ^ 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= round)
                __builtin_amdgcn_s_sleep(1);
        }
        if (recv) {
            const int j = wave ? g - (n + 1 - kLanes) : g - split;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                L.x[0][c] = ex[b][1 - wave][c][j];
                L.v[0][c] = ex[b][1 - wave][3 + c][j];
                L.w[0][c] = ex[b][1 - wave][6 + c][j];
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) L.Q[0][c] = ex[b][1 - wave][9 + c][j];
        }
        if constexpr (RPB == 1) __syncthreads();
        ++round;
    };

    // an env that already holds a NaN is not integrated (see softrod_step_fast_kernel)
    bool dead = false;
    {
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            bad = bad || (owned && (isnan(L.x[0][c]) || isnan(L.v[0][c]) || (g < n && isnan(L.w[0][c]))));
#pragma unroll
        for (int c = 0; c < 9; ++c) bad = bad || (owned && g < n && isnan(L.Q[0][c]));
        if constexpr (RPB == 1) dead = __syncthreads_or(bad ? 1 : 0) != 0;
        else {      // per rod: both of its waves vote through LDS; every wave of the block takes the barriers
            __syncthreads();                                   // counters and votes cleared
            if (__any(bad) && lane == 0) atomicOr(&sany_[es], 1);
            __syncthreads();
            dead = sany_[es] != 0;
        }
    }
    if (!live) return;                       // (RPB > 1; no barrier below)
    if (dead) {
        poison_rod<1>(L);
    } else if (n_sub > 0) {
        kinematic_n<1>(P.half_dt, C, L);
        int since = 0;
        for (int s = 0; s < n_sub; ++s) {
#if SOFTROD_WINDOW_PRIO
            // The two windows of a rod share a SIMD (RPB = 4).  The arbiter serves the older wave
            // first, so one window would run ahead to the rendezvous and sleep while the other
            // finishes ALONE, at a lone wave's issue rate.  A wave's priority falls as it advances
            // through the refresh interval, so whoever is behind is served first and the two
            // arrive together (ProgressPriority's idea, softrod_fast.hpp).
            if constexpr (RPB > 1) {
                if (since == 0) __builtin_amdgcn_s_setprio(2);
                else if (since == 1) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
#endif
            dynamic_n<F, 1>(Pk, C, B, gi, L);
            const bool last = (s == n_sub - 1);
            kinematic_n<1>(last ? P.half_dt : P.dt, C, L);
            if (++since == refresh) { exchange(); since = 0; }
        }
    }
    time = clock_after(P, S, time, n_sub);
    if (owned) {
        window_store(S, N, rod, g, L);
        constexpr size_t W = 2 * kLanes;
        if (actions) S.rkap[(size_t)rod * W + g] = L.rk[0][0];
    }
    if (tid == 0) S.time[rod] = time;
}

}  // namespace softrod
