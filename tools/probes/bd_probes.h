/* C ABI of the diagnostic probe library (tools/probes/libbd_probes.so; `make -C tools/probes`).  NOT part of the product
 * library: experiments that informed the kernel design and are kept runnable. */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
/* ---- weight-stationary dense layer for tall inputs (csrc/dense_ws.hip): one layer of a DenseModel
 * (src/models.py:365-408) over M >> N rows with the weights resident in registers for the whole launch.
 * Forward (saved == NULL): out[M x N] = act(in[M x K] W^T + bias), w_packed = packed (N, K).
 * Backward dgrad (saved != NULL): out = (in W^T) * ELU'(saved[M x N]) -- with in = the gradient w.r.t. a layer's
 * pre-activation and w_packed its packed TRANSPOSE, out is the gradient w.r.t. the previous layer's pre-activation.
 * bd_dense_ws_supported: shapes this kernel takes (today: 193 <= K <= 208, K % 4 == 0, N <= 256, M >= 16). */
int bd_dense_ws_supported(int M, int N, int K);
int bd_dense_ws(const float* in, int ldi, const float* w_packed, const float* bias, const float* saved, int M, int N, int K,
                int act, float* out, int ldo, void* stream);
/* Diagnostic: register-only v_mfma_f32_16x16x4_f32 loop (16 waves per workgroup, 4 independent chains per wave) to measure
 * the fp32 MFMA rate this part sustains; flops = blocks * 16 * iters * 32 * 2048. */
int bd_mfma_probe(int blocks, int iters, float* out, void* stream);

const char* bd_probe_last_error(void);
#ifdef __cplusplus
}
#endif
