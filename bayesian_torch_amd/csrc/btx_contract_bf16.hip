// Instantiates the bf16-MFMA (throughput mode) variants of the fused contraction: v_mfma_f32_32x32x16_bf16.
#include "btx_contract.h"
#include "btx_contract_dma.h"
#include "btx_contract_gemm8.h"
#if defined(BTX_TUNING) || defined(BTX_PT_TRACE)
#include "../../tools/experimental/btx_contract_pw.h"  // measured and parked: see btx_api.hip
#endif
namespace btx {
int launch_contract_bf16(int kind, int act_bf16, bool gen, const ContractParams& p, int nwg, hipStream_t st) {
  return launch_contract_impl<1>(kind, act_bf16, gen, p, nwg, st);
}
int launch_contract_dma_bf16(int kind, const ContractParams& p, int nwg, hipStream_t st) {
  return launch_contract_dma_impl<1>(kind, p, nwg, st);
}
#if defined(BTX_TUNING) || defined(BTX_PT_TRACE)
int launch_contract_pw_bf16(int kind, const ContractParams& p, int nwg, hipStream_t st) {
  return launch_contract_pw_impl<1>(kind, p, nwg, st);
}
#endif
int launch_contract_gemm8_bf16(int kind, const ContractParams& p, int nwg, hipStream_t st) {
  return launch_contract_gemm8_impl<1>(kind, p, nwg, st);
}
}  // namespace btx
