// Issue cost of the instruction kinds the rod kernels are made of, measured on the device:
// cycles per wave64 instruction with 1, 2, 4 waves per SIMD, independent and dependent chains.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue tools/microbench/valu_issue.hip && ./valu_issue
// The shader clock (s_memtime) brackets REP x 64 instructions per wave; the figure printed is
// (max over the waves of a SIMD-filling launch) / (REP x 64 x waves per SIMD): 4.0 means one wave64
// instruction per 4 cycles per SIMD, the rate every roofline number in DESIGN.md assumes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <map>

#define REP 512

#define X4(s) s s s s
#define X16(s) X4(s) X4(s) X4(s) X4(s)
#define X64(s) X16(s) X16(s) X16(s) X16(s)

enum Kind { RSQ32_IND = 100, CVT_F32_F64, CVT_F64_F32, RSQ_VIA_F32, FMA_IND, FMA_DEP, MUL_IND, ADD_IND, FMAC_IND, DPP_IND, DPP_DEP, RSQ_IND, RCP_IND, FMA32_IND, DPP_THEN_FMA, MIX };

template <int K>
__global__ void __launch_bounds__(64) bench(unsigned long long* out, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double m = 1.0000001, c = 1e-9;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(m), "+v"(c));
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long r0 = wall_clock64();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
        if (K == FMA_IND) {
            asm volatile(X16("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (K == FMA_DEP) {
            asm volatile(X64("v_fma_f64 %0, %0, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (K == MUL_IND) {
            asm volatile(X16("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (K == ADD_IND) {
            asm volatile(X16("v_add_f64 %0, %0, %9\n v_add_f64 %1, %1, %9\n v_add_f64 %2, %2, %9\n v_add_f64 %3, %3, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (K == FMAC_IND) {
            asm volatile(X16("v_fmac_f64_e32 %0, %8, %9\n v_fmac_f64_e32 %1, %8, %9\n v_fmac_f64_e32 %2, %8, %9\n v_fmac_f64_e32 %3, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (K == DPP_IND) {
            asm volatile(X16("v_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
        } else if (K == DPP_DEP) {
            asm volatile(X64("s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(i0) :);
        } else if (K == RSQ_IND) {
            asm volatile(X16("v_rsq_f64 %0, %4\n v_rsq_f64 %1, %5\n v_rsq_f64 %2, %6\n v_rsq_f64 %3, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        } else if (K == RCP_IND) {
            asm volatile(X16("v_rcp_f64 %0, %4\n v_rcp_f64 %1, %5\n v_rcp_f64 %2, %6\n v_rcp_f64 %3, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        } else if (K == RSQ32_IND) {
            asm volatile(X16("v_rsq_f32 %0, %4\n v_rsq_f32 %1, %5\n v_rsq_f32 %2, %6\n v_rsq_f32 %3, %7\n")
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));
        } else if (K == CVT_F32_F64) {
            asm volatile(X16("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n")
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        } else if (K == CVT_F64_F32) {
            asm volatile(X16("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
        } else if (K == RSQ_VIA_F32) {   // the 2^-23 seed of 1/sqrt(x) by way of float32: 3 instructions, counted as one
            asm volatile(X16("v_cvt_f32_f64 %4, %8\n v_rsq_f32 %4, %4\n v_cvt_f64_f32 %0, %4\n v_cvt_f32_f64 %5, %9\n v_rsq_f32 %5, %5\n v_cvt_f64_f32 %1, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a4), "v"(a5));
        } else if (K == FMA32_IND) {
            asm volatile(X16("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(1.0000001f), "v"(1e-9f));
        } else if (K == DPP_THEN_FMA) {   // the rod loop's pattern: two DPP moves feed one fp64 operation
            asm volatile(X16("v_mov_b32_dpp %4, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_fma_f64 %2, %2, %6, %7\n v_fma_f64 %3, %3, %6, %7\n")
                         : "+v"(i0), "+v"(i1), "+v"(a2), "+v"(a3), "+v"(i2), "+v"(i3) : "v"(m), "v"(c));
        } else if (K == MIX) {            // 2 fma : 1 mul : 1 add, independent
            asm volatile(X16("v_fma_f64 %0, %0, %8, %9\n v_mul_f64 %1, %1, %8\n v_fma_f64 %2, %2, %8, %9\n v_add_f64 %3, %3, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    asm volatile("" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(i0), "v"(i1), "v"(i2), "v"(i3));
    if (threadIdx.x == 0) {
        out[blockIdx.x] = t1 - t0;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        // HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx9); XCC_ID 3:0
        out[8192 + blockIdx.x] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
        out[16384 + blockIdx.x] = r0;
        out[24576 + blockIdx.x] = r1;
    }
}

template <int K>
void run(const char* name, unsigned long long* d_out) {
    printf("%-34s", name);
    for (int wps : {1, 2, 4, 8}) {
        const int waves = 1024 * wps;          // 256 CUs x 4 SIMDs
        std::vector<unsigned long long> h(waves), id(waves), st(waves), en(waves);
        for (int rep = 0; rep < 3; ++rep) bench<K><<<waves, 64>>>(d_out, 1.0);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_out, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        hipMemcpy(id.data(), d_out + 8192, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        hipMemcpy(st.data(), d_out + 16384, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        hipMemcpy(en.data(), d_out + 24576, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        // 100 MHz wall clock: the launch's span, a wave's own span, hence how many waves of a SIMD ran at once,
        // and the rate of the s_memtime counter
        unsigned long long first = *std::min_element(st.begin(), st.end()), last = *std::max_element(en.begin(), en.end());
        double own = 0, rate = 0;
        for (int w = 0; w < waves; ++w) { own += (double)(en[w] - st[w]); rate += (double)h[w] / (double)(en[w] - st[w]) / 10.0; }
        own /= waves; rate /= waves;
        const double span_us = (last - first) / 100.0, own_us = own / 100.0;
        // waves per SIMD that START within the first wave's duration of the earliest start (co-resident)
        std::map<unsigned long long, int> per_simd;       // key: everything of HW_ID but the wave slot, + XCC
        for (int w = 0; w < waves; ++w) per_simd[id[w] & ~0xfull]++;

        int mn = 1 << 30, mxs = 0;
        for (auto& kv : per_simd) { mn = std::min(mn, kv.second); mxs = std::max(mxs, kv.second); }
        std::sort(h.begin(), h.end());
        const double med = (double)h[waves / 2];
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int rep = 0; rep < 4; ++rep) bench<K><<<waves, 64>>>(d_out, 1.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ns_per_simd_instr = ms * 1e6 / 4.0 / (REP * 64.0 * wps);
        (void)ns_per_simd_instr; (void)mn; (void)mxs;
        printf(" | %dw: %5.2f cyc %5.2f ns/wave-instr, span %4.0f us, conc %.1f, %.2f GHz", wps, med / (REP * 64.0),
               own_us * 1e3 / (REP * 64.0), span_us, wps * own_us / span_us, rate);
    }
    printf("\n");
}

int main() {
    unsigned long long* d_out;
    hipMalloc(&d_out, 4 * 8192 * sizeof(unsigned long long));
    printf("per launch of 1024 x w one-wave workgroups (w waves on every SIMD): shader cycles (s_memtime) and ns (100 MHz wall clock) per wave64 instruction of ONE wave, the launch's span, the average number of waves of a SIMD running at once, the shader clock\n");
    run<FMA_IND>("v_fma_f64 independent x4", d_out);
    run<FMA_DEP>("v_fma_f64 dependent chain", d_out);
    run<MUL_IND>("v_mul_f64 independent", d_out);
    run<ADD_IND>("v_add_f64 independent", d_out);
    run<FMAC_IND>("v_fmac_f64 independent", d_out);
    run<MIX>("fma/mul/fma/add independent", d_out);
    run<DPP_IND>("v_mov_b32_dpp wave_shr independent", d_out);
    run<DPP_DEP>("v_mov_b32_dpp wave_shr dependent", d_out);
    run<DPP_THEN_FMA>("2 dpp + 2 fma_f64", d_out);
    run<RSQ_IND>("v_rsq_f64 independent", d_out);
    run<RCP_IND>("v_rcp_f64 independent", d_out);
    run<FMA32_IND>("v_fma_f32 independent", d_out);
    run<RSQ32_IND>("v_rsq_f32 independent", d_out);
    run<CVT_F32_F64>("v_cvt_f32_f64 independent", d_out);
    run<CVT_F64_F32>("v_cvt_f64_f32 independent", d_out);
    run<RSQ_VIA_F32>("cvt,rsq_f32,cvt (per 1.5 instr)", d_out);
    // what the counter's tick is: compare with the wall clock
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) bench<FMA_IND><<<4096, 64>>>(d_out, 1.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(4096);
        hipMemcpy(h.data(), d_out, 4096 * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("20 launches of 4096 waves: %.3f ms wall; per launch %.1f us; median ticks per wave %.0f -> %.1f ticks/us\n",
               ms, ms * 50.0, (double)h[2048], (double)h[2048] / (ms * 50.0));
    }
    return 0;
}
