// bd_cluster.h -- what the CLUSTER scans share (observe_cluster.hip: Gaussian latents; observe_cat_cluster.hip: Categorical
// latents): several workgroups (one per CU) own one 16-row tile of a recurrent scan and exchange one or two small vectors
// per time step through a buffer in L2.
//
// Hand-off protocol (cdna_hip_programming.md, Guideline 16 form R1 / MI355X_MICROARCH.md "Valid forms"):
//   producer: payload stored write-through (sc1: relaxed agent-scope atomic stores) -> every storing wave
//             s_waitcnt vmcnt(0) -> workgroup barrier -> ONE lane stores the member's flag (sc1) = epoch;
//   consumer: ONE wave polls the C flags with relaxed agent-scope loads (lane i polls member i) until all have
//             reached the epoch -> workgroup barrier -> EVERY load of the payload is an sc1 load to registers.
//   Epochs are monotonic within a launch and never 0; flags are zeroed by a memset node ahead of the launch.
//   Every spin is bounded: on timeout the member ORs its code into the STICKY error word and stops waiting (outputs are
//   then wrong, the launch still terminates).  The error word is NOT part of the per-launch header memset: it survives
//   later launches until bd_observe_cluster_status reads (and clears) it -- the engine reads it with every log fetch.
// Workspace header (floats): [flags: tiles * kMaxCluster u32][err: 16 u32, sticky]; payload layouts belong to the kernels.
#pragma once
#include "bd_device.h"
#include "bd_host.h"
#include <stdlib.h>

namespace bd {

constexpr int kMaxCluster = 16;

constexpr int kLocalBlocks = 2;             // column blocks a member owns at most (host picks C accordingly)
constexpr unsigned kSpinLimit = 1u << 22;   // ~ seconds; far beyond any legitimate wait
// host copy of the spin limit, passed to every launch (bd_observe_cluster_set_spin_limit: tests); one per process
unsigned& cluster_spin_limit();      // observe_cluster.hip
constexpr unsigned kErrFwd = 1u, kErrBwd = 2u;


// workspace: [flags: tiles*kMaxCluster u32][err: 16 u32, sticky][payload: tiles * 2 parities * nvec * Kb_h*256 floats]
__host__ __device__ inline size_t cluster_ws_flag_floats(int tiles) { return (size_t)tiles * kMaxCluster; }
__host__ __device__ inline size_t cluster_ws_header_floats(int tiles) { return cluster_ws_flag_floats(tiles) + 16; }

__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_sc1_u64(const float* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// publish this member's payload (already stored with st_sc1 by `storing` waves) under `epoch`
__device__ __forceinline__ void publish(unsigned* flag, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave: its sc1 stores have left
    lds_barrier();
    if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave 0 polls the C member flags of this tile; returns after a workgroup barrier
__device__ __forceinline__ void wait_all(const unsigned* flags, int C, unsigned epoch, unsigned* err, unsigned limit,
                                         unsigned code) {
    if ((threadIdx.x >> 6) == 0) {
        const int lane = threadIdx.x & 63;
        unsigned spins = 0;
        for (;;) {
            unsigned v = epoch;
            if (lane < C) v = __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(v >= epoch)) break;
            if (++spins > limit) {
                if (lane == 0) __hip_atomic_fetch_or(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    lds_barrier();
}

// copy `nfl` floats (multiple of 2) from the exchange buffer into LDS with sc1 loads
__device__ __forceinline__ void gather_payload(const float* __restrict__ src, float* __restrict__ dst, int nfl) {
    for (int i = threadIdx.x * 2; i < nfl; i += blockDim.x * 2) {
        const unsigned long long u = ld_sc1_u64(src + i);
        dst[i] = __uint_as_float((unsigned)u);
        dst[i + 1] = __uint_as_float((unsigned)(u >> 32));
    }
}

// The scan's members advance in lock step (one all-gather per time step), so a member that shares its CU's issue
// slots with another stream's workgroups slows the whole cluster (measured under the engine's cross-step pipeline:
// observe_bwd 1.0 -> 1.75 ms).  Requesting the CU's whole LDS keeps every LDS-using kernel of the other streams off
// a member's CU.  BD_OBS_EXCLUSIVE=0 launches with the LDS the kernel needs.
// A workgroup's LDS allocation is static + dynamic: the dynamic request is what is left of the CU's LDS after the
// kernel's static __shared__ (0 in the product build; a diagnostic build that adds any would otherwise ask for more than
// the CU has and the queue aborts with HSA_STATUS_ERROR_INVALID_ALLOCATION instead of returning an error).
// Returns 0 and sets the error text when `need` does not fit.
template <class K>
inline size_t launch_lds(K kernel, size_t need, const char* who) {
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, reinterpret_cast<const void*>(kernel)) != hipSuccess) {
        fail("%s: hipFuncGetAttributes failed", who);
        return 0;
    }
    const size_t room = (size_t)kMaxLds > at.sharedSizeBytes ? (size_t)kMaxLds - at.sharedSizeBytes : 0;
    if (need > room) {
        fail("%s: needs %zu B of dynamic LDS, %zu B available beside %zu B of static __shared__", who, need, room,
             (size_t)at.sharedSizeBytes);
        return 0;
    }
    static const char* e = getenv("BD_OBS_EXCLUSIVE");
    return (e && e[0] == '0') ? need : room;
}

// K-split form of the Gaussian cluster scan (observe_ksplit.hip): used by bd_observe_forward_cluster / _backward_cluster when
// the cluster has one member per 16-column belief block (ksplit_ok); BD_OBS_KSPLIT=0 keeps the round-1 form
size_t ksplit_ws_floats_per_tile(int C);
bool ksplit_ok(int Be, int S, int A, int Hd, int C);
int& ksplit_mode();
int launch_observe_kfwd(const bd_observe_fwd_args* a, float* ws, int C, int tiles, hipStream_t stream);
int launch_observe_kbwd(const bd_observe_bwd_args* a, float* ws, int C, int tiles, hipStream_t stream);

}  // namespace bd
