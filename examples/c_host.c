/*
 * Minimal C host of the C-ABI (include/softrod.h): no Python, no torch — the HIP runtime
 * for device buffers, libsoftrod_hip.so for everything else.  Steps N SoftPendulum-v0 envs
 * and prints each env's observation, reward and flags after every env.step, in a format
 * tests/test_gpu_c_host.py compares with the Python host path.
 *
 *   gcc -std=gnu11 -o c_host examples/c_host.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
 *       -Lgym_softrobot_amd/csrc -lsoftrod_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/gym_softrobot_amd/csrc -Wl,-rpath,/opt/rocm/lib -lm
 *   ./c_host 4 3          # 4 envs, 3 steps
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "softrod.h"

#define CHECK(call)                                                                    \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ != 0) {                                                                \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, softrod_last_error(h));      \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 4, steps = argc > 2 ? atoi(argv[2]) : 3;
    softrod_handle* h = NULL;
    softrod_config cfg;
    CHECK(softrod_config_softpendulum(&cfg, n));
    CHECK(softrod_create(&cfg, 0, &h));

    /* build.py:47-49 draws theta0 from the env's RNG; here a fixed table */
    double* theta0 = (double*)malloc(sizeof(double) * n);
    for (int i = 0; i < n; ++i) theta0[i] = (90.0 + (0.1 * i - 0.2) * 10.0) * M_PI / 180.0;
    CHECK(softrod_reset(h, theta0, NULL, NULL));

    float *d_act, *d_obs, *h_obs = (float*)malloc(sizeof(float) * 4 * n), *h_act = (float*)malloc(sizeof(float) * n);
    double *d_rew, *h_rew = (double*)malloc(sizeof(double) * n);
    uint8_t *d_term, *d_trunc, *h_flag = (uint8_t*)malloc(2 * n);
    if (hipMalloc((void**)&d_act, sizeof(float) * n) || hipMalloc((void**)&d_obs, sizeof(float) * 4 * n) ||
        hipMalloc((void**)&d_rew, sizeof(double) * n) || hipMalloc((void**)&d_term, n) || hipMalloc((void**)&d_trunc, n))
        return 2;
    CHECK(softrod_observe(h, NULL, d_obs, NULL));
    hipMemcpy(h_obs, d_obs, sizeof(float) * 4 * n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i)
        printf("reset %d %.9g %.9g %.9g %.9g\n", i, h_obs[4 * i], h_obs[4 * i + 1], h_obs[4 * i + 2], h_obs[4 * i + 3]);
    for (int t = 0; t < steps; ++t) {
        for (int i = 0; i < n; ++i) h_act[i] = (float)(22.0 * sin(1.0 + 0.7 * i + 1.3 * t));
        hipMemcpy(d_act, h_act, sizeof(float) * n, hipMemcpyHostToDevice);
        CHECK(softrod_step(h, d_act, d_obs, d_rew, d_term, d_trunc, NULL, NULL));
        hipMemcpy(h_obs, d_obs, sizeof(float) * 4 * n, hipMemcpyDeviceToHost);   /* syncs the null stream */
        hipMemcpy(h_rew, d_rew, sizeof(double) * n, hipMemcpyDeviceToHost);
        hipMemcpy(h_flag, d_term, n, hipMemcpyDeviceToHost);
        hipMemcpy(h_flag + n, d_trunc, n, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i)
            printf("step %d %d %.9g %.9g %.9g %.9g %.17g %d %d\n", t, i, h_obs[4 * i], h_obs[4 * i + 1],
                   h_obs[4 * i + 2], h_obs[4 * i + 3], h_rew[i], h_flag[i], h_flag[n + i]);
    }
    CHECK(softrod_destroy(h));
    return 0;
}
