// Instantiates the split-bf16 ("bf16x3", BTX_PREC_BF16X3) variants of the LDS-DMA kernel family: f32 activations and the
// f32 modes' LDS images, three v_mfma_f32_32x32x16_bf16 per product (btx_mma.h).  Shapes outside this family run on the
// exact-f32 kernels of btx_contract_f32.hip (at least as accurate).
#include "btx_contract_patch.h"
#include "btx_contract_stem.h"
#include "btx_contract_dma.h"
#include "btx_contract_gemm8.h"
#if defined(BTX_TUNING) || defined(BTX_PT_TRACE)
#include "../../tools/experimental/btx_contract_pw.h"  // measured and parked: see btx_api.hip
#endif
namespace btx {
int launch_contract_patch_x3(int kind, const ContractParams& p, int nwg, hipStream_t st) {
  return launch_contract_patch_impl<2>(kind, p, nwg, st);
}
int launch_contract_stem_x3(int kind, const ContractParams& p, int nwg, hipStream_t st) {
  return launch_contract_stem_impl<2>(kind, p, nwg, st);
}
int launch_contract_dma_x3(int kind, const ContractParams& p, int nwg, hipStream_t st) {
  return launch_contract_dma_impl<2>(kind, p, nwg, st);
}
#if defined(BTX_TUNING) || defined(BTX_PT_TRACE)
int launch_contract_pw_x3(int kind, const ContractParams& p, int nwg, hipStream_t st) {
  return launch_contract_pw_impl<2>(kind, p, nwg, st);
}
#endif
int launch_presample_batch_x3(const PresampleBatch& b, hipStream_t st) {
  hipLaunchKernelGGL((presample_batch_kernel<2>), dim3(b.total_blocks), dim3(256), 0, st, b);
  return (int)hipGetLastError();
}
int launch_contract_gemm8_x3(int kind, const ContractParams& p, int nwg, hipStream_t st) {
  return launch_contract_gemm8_impl<2>(kind, p, nwg, st);
}
}  // namespace btx
