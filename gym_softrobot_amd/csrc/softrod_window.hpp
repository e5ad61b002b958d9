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

// grid = n_envs, block = 128.  ArmSingle feature set (no boundary condition, no filter).
template <unsigned F>
__global__ void __launch_bounds__(2 * kLanes, SOFTROD_CONTACT_WAVES)
softrod_step_window_kernel(const RodParams P, const StatePtrs S, const float* __restrict__ actions,
                           const int n_sub, const int refresh) {
    constexpr int kFields = 18;
    __shared__ double ex[2][kFields][kLanes];

    const int rod = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t N = (size_t)P.n_envs;
    const int n = P.n_elem;
    if (S.skip && S.skip[rod]) return;            // reset by the auto-reset pass; the epilogue launch clears the flag
    const int off = wave ? (n + 1 - kLanes) : 0;   // first node of this wave's window
    const int g = off + lane;                      // this lane's node / element / Voronoi index
    const int split = (n + 1) / 2;                 // wave 0 owns nodes < split, wave 1 the rest
    const bool owned = wave ? (g >= split) : (g < split);
    // The last lane of wave 0 is an interior node whose "next" shifts in zeros.  It is given the
    // index of the rod's end node, so that its element / Voronoi vertex are inert (no stiffness,
    // excluded from the wave-uniform range checks) instead of producing an absurd strain.
    const int gi = (wave == 0 && lane == kLanes - 1) ? n : g;

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
    auto exchange = [&]() {
        if (send) {
            const int j = wave ? g - split : g - (n + 1 - kLanes);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                ex[wave][c][j] = L.x[0][c];
                ex[wave][3 + c][j] = L.v[0][c];
                ex[wave][6 + c][j] = L.w[0][c];
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) ex[wave][9 + c][j] = L.Q[0][c];
        }
        __syncthreads();
        if (recv) {
            const int j = wave ? g - (n + 1 - kLanes) : g - split;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                L.x[0][c] = ex[1 - wave][c][j];
                L.v[0][c] = ex[1 - wave][3 + c][j];
                L.w[0][c] = ex[1 - wave][6 + c][j];
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) L.Q[0][c] = ex[1 - wave][9 + c][j];
        }
        __syncthreads();
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
        dead = __syncthreads_or(bad ? 1 : 0) != 0;
    }
    if (dead) {
        poison_rod<1>(L);
    } else if (n_sub > 0) {
        kinematic_n<1>(P.half_dt, C, L);
        int since = 0;
        for (int s = 0; s < n_sub; ++s) {
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
