// issue_probe.hip -- how do MFMA streams and other instructions share a SIMD on gfx950?
//  A (inter-wave): one workgroup of 8 waves (w and w+4 share a SIMD).  Wave 0 streams v_mfma_f32_16x16x4_f32 -- either
//    back to back, or in runs of 36 separated by six ds_read_b128 + s_waitcnt like a real sweep -- while wave 4 times a
//    section of VALU / LDS / global-store instructions with s_memtime.
//  B (intra-wave): a single wave issues 1 MFMA + k independent v_add_f32 per iteration: do the adds hide under the MFMA?
// Build and run:  hipcc --offload-arch=gfx950 -O3 tools/probes/issue_probe.hip -o tools/probes/issue_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void probe(int mode, int with_mfma, int prio, unsigned long long* out, float* sink) {
    __shared__ floatx4 lds4[1024];
    float* lds = reinterpret_cast<float*>(lds4);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {       // MFMA streamer
        if (!with_mfma) return;
        floatx4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = lane * 0.001f, y = 1.0f;
        if (with_mfma == 1) {
            for (int i = 0; i < 4000; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            }
        } else {
            for (int i = 0; i < 444; ++i) {     // runs of 36 MFMAs, then 6 LDS reads that the next run depends on
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
                }
                floatx4 s = {0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 6; ++j) s += lds4[(lane + 64 * j + i) & 1023];
                x += s[0] * 1e-30f;
                y += s[1] * 1e-30f;
            }
        }
        sink[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
        return;
    }
    if (wave != 4) return;
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(10);    // let the streamer get going
    if (prio) __builtin_amdgcn_s_setprio(3);
    float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {            // 1024 VALU adds (4 chains)
#pragma unroll 1
        for (int i = 0; i < 256; ++i) {
            asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        }
    } else if (mode == 1) {     // 256 LDS writes
#pragma unroll 1
        for (int i = 0; i < 256; ++i) {
            lds[(lane + i * 64) & 4095] = v0;
            asm volatile("" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (mode == 2) {     // 256 global stores (dword per lane, coalesced)
#pragma unroll 1
        for (int i = 0; i < 256; ++i) {
            sink[1024 + lane + i * 64] = v0;
            asm volatile("" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[0] = t1 - t0;
    sink[512 + threadIdx.x] = v0 + v1 + v2 + v3;
}

template <int K>
__global__ __launch_bounds__(64) void intra(unsigned long long* out, float* sink) {
    const int lane = threadIdx.x;
    floatx4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = lane * 0.001f, y = 1.0f;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = lane + k;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 1000; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[k & 7]));
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[k & 7]));
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[k & 7]));
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[k & 7]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[0] = t1 - t0;
    float s = a0[0] + a1[1] + a2[2] + a3[3];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
    sink[lane] = s;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned long long* out;
    float* sink;
    hipMalloc(&out, 64);
    hipMalloc(&sink, 1 << 20);
    printf("issue_probe: device ready\n");
    const char* names[] = {"1024 v_add_f32", "256 ds_write_b32", "256 global_store_dword"};
    const char* sn[] = {"none", "back-to-back MFMAs", "runs of 36 + 6 ds_read"};
    for (int mode = 0; mode < 3; ++mode)
        for (int with = 0; with < 3; ++with)
            for (int prio = 0; prio < 2; ++prio) {
                if (with == 0 && prio) continue;
                unsigned long long h = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, mode, with, prio, out, sink);
                    hipDeviceSynchronize();
                }
                hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
                printf("A  %-24s SIMD-mate: %-24s prio %d : %8llu cycles\n", names[mode], sn[with], prio, h);
            }
    unsigned long long h = 0;
#define RUN_INTRA(K)                                                     \
    for (int rep = 0; rep < 2; ++rep) {                                  \
        hipLaunchKernelGGL(intra<K>, dim3(1), dim3(64), 0, 0, out, sink); \
        hipDeviceSynchronize();                                          \
    }                                                                    \
    hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);                         \
    printf("B  one wave, 4000 x (1 MFMA + %d v_add_f32): %8llu cycles = %.1f per MFMA\n", K, h, h / 4000.0);
    RUN_INTRA(0) RUN_INTRA(1) RUN_INTRA(2) RUN_INTRA(4) RUN_INTRA(6) RUN_INTRA(8) RUN_INTRA(12)
    return 0;
}
