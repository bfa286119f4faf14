// cabi.hip -- error reporting, version, weight packing, replay gather.
#include "bd_device.h"
#include "bd_host.h"
#include "bd_rng.h"

namespace bd {
char* err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}

// One block column per descriptor (blockIdx.y); blockIdx.x strides over its 16x16 blocks; each thread
// writes one packed float4: dst[(nb*Kb + kb)*64 + lane] = {W[n][k0..k0+3]}, n = nb*16 + (lane&15),
// k0 = kb*16 + 4*(lane>>4); transposed descriptors read W[k][n] instead.
__global__ __launch_bounds__(256) void pack_kernel(const bd_pack_desc* __restrict__ descs) {
    const bd_pack_desc d = descs[blockIdx.y];
    const int No = d.transpose ? d.K : d.N;   // packed "out" dim
    const int Ki = d.transpose ? d.N : d.K;   // packed "in" dim
    const int Nb = (No + 15) >> 4, Kb = (Ki + 15) >> 4;
    const int total = Nb * Kb * 64;
    floatx4* __restrict__ dst = reinterpret_cast<floatx4*>(d.dst);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int lane = e & 63, blk = e >> 6;
        const int nb = blk / Kb, kb = blk - nb * Kb;
        const int n = nb * 16 + (lane & 15);
        const int k0 = kb * 16 + 4 * (lane >> 4);
        floatx4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + i;
            float x = 0.f;
            if (n < No && k < Ki) x = d.transpose ? d.src[(size_t)k * d.ld + n] : d.src[(size_t)n * d.ld + k];
            v[i] = x;
        }
        dst[e] = v;
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                          int n_idx, int width, float* __restrict__ dst) {
    const size_t total = (size_t)n_idx * width;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / width, c = e - r * width;
        dst[e] = src[(size_t)idx[r] * width + c];
    }
}
// Pixel replay: gather uint8 frames by row index and undo the bit-depth quantisation in one pass
// (ExperienceReplay._retrieve_batch + preprocess_observation_, src/memory.py:70-85, src/utils.py:299-317):
//   out = floor(u8 / 2^(8-bits)) / 2^bits - 0.5 + noise / 2^bits        (noise ~ U[0,1), explicit input)
// HBM-bound byte work: 4 pixels per thread (uchar4 in, float4 noise in, float4 out), fully coalesced.
// RNG = true (perf mode): the dequantisation noise U[0, 1) of the four pixels is one Philox4x32-10 call (bd_rng.h) instead of
// a 16-byte read of a noise tensor that a library kernel wrote (120 MB per step at configs[2], written and read once).
template <bool RNG>
__global__ __launch_bounds__(256) void gather_pixels_kernel(const unsigned char* __restrict__ src,
                                                            const int64_t* __restrict__ idx, int n_idx, int pixels,
                                                            float inv_q, float inv_b, const float* __restrict__ noise,
                                                            float* __restrict__ dst, Rng rng) {
    const int quads = pixels >> 2;
    const size_t total = (size_t)n_idx * quads;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / quads, q = e - r * quads;
        const uchar4 u = reinterpret_cast<const uchar4*>(src + (size_t)idx[r] * pixels)[q];
        floatx4 nz;
        if constexpr (RNG) {
            const Philox4 p = philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), rng.stream, rng.step, rng.k0, rng.k1);
#pragma unroll
            for (int j = 0; j < 4; ++j) nz[j] = (float)(p.x[j] >> 8) * (1.0f / 16777216.0f);       // [0, 1) as rand_like
        } else {
            nz = reinterpret_cast<const floatx4*>(noise)[e];
        }
        floatx4 o;
        o[0] = floorf((float)u.x * inv_q) * inv_b - 0.5f + nz[0] * inv_b;
        o[1] = floorf((float)u.y * inv_q) * inv_b - 0.5f + nz[1] * inv_b;
        o[2] = floorf((float)u.z * inv_q) * inv_b - 0.5f + nz[2] * inv_b;
        o[3] = floorf((float)u.w * inv_q) * inv_b - 0.5f + nz[3] * inv_b;
        reinterpret_cast<floatx4*>(dst)[e] = o;
    }
}
}  // namespace bd

extern "C" {

const char* bd_last_error(void) { return bd::err_buf(); }
int bd_version(void) { return 1; }

size_t bd_packed_floats(int N, int K) { return (size_t)bd::cdiv(N, 16) * bd::cdiv(K, 16) * 256; }

int bd_pack_weights(const bd_pack_desc* descs, int n, void* stream) {
    BD_REQUIRE(descs != nullptr && n > 0, "bd_pack_weights: no descriptors");
    hipLaunchKernelGGL(bd::pack_kernel, dim3(16, n), dim3(256), 0, (hipStream_t)stream, descs);
    BD_CHECK_LAUNCH("bd_pack_weights");
    return 0;
}

int bd_replay_gather(const float* src, const int64_t* idx, int n_idx, int width, float* dst, void* stream) {
    BD_REQUIRE(src && idx && dst && n_idx > 0 && width > 0, "bd_replay_gather: bad arguments");
    const size_t total = (size_t)n_idx * width;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(bd::gather_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, idx, n_idx, width, dst);
    BD_CHECK_LAUNCH("bd_replay_gather");
    return 0;
}

int bd_replay_gather_pixels(const unsigned char* src, const int64_t* idx, int n_idx, int pixels, int bit_depth,
                            const float* noise, float* dst, void* stream) {
    BD_REQUIRE(src && idx && noise && dst && n_idx > 0 && pixels > 0 && (pixels & 3) == 0 && bit_depth >= 1 && bit_depth <= 8,
               "bd_replay_gather_pixels: bad arguments");
    const size_t total = (size_t)n_idx * (pixels >> 2);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(bd::gather_pixels_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, idx, n_idx, pixels,
                       1.0f / (float)(1 << (8 - bit_depth)), 1.0f / (float)(1 << bit_depth), noise, dst, bd::Rng{0, 0, 0, 0});
    BD_CHECK_LAUNCH("bd_replay_gather_pixels");
    return 0;
}

int bd_replay_gather_pixels_rng(const unsigned char* src, const int64_t* idx, int n_idx, int pixels, int bit_depth,
                                unsigned long long seed, unsigned long long step, float* dst, void* stream) {
    BD_REQUIRE(src && idx && dst && n_idx > 0 && pixels > 0 && (pixels & 3) == 0 && bit_depth >= 1 && bit_depth <= 8,
               "bd_replay_gather_pixels_rng: bad arguments");
    const size_t total = (size_t)n_idx * (pixels >> 2);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(bd::gather_pixels_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, idx, n_idx, pixels,
                       1.0f / (float)(1 << (8 - bit_depth)), 1.0f / (float)(1 << bit_depth), nullptr, dst,
                       bd::Rng{(uint32_t)seed, (uint32_t)(seed >> 32), 6u, (uint32_t)step});
    BD_CHECK_LAUNCH("bd_replay_gather_pixels_rng");
    return 0;
}

}  // extern "C"
