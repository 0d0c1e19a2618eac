// Instantiates the bf16 "patch" variant (stride-1 2-D convolutions, halo'd input patch resident in LDS).
#include "btx_contract_patch.h"
#include "btx_contract_stem.h"
#include "btx_contract_stempool.h"
namespace btx {
int launch_contract_patch_bf16(int kind, const ContractParams& p, int nwg, hipStream_t st) {
  return launch_contract_patch_impl<1>(kind, p, nwg, st);
}
int launch_contract_stem_bf16(int kind, const ContractParams& p, int nwg, hipStream_t st) {
  return launch_contract_stem_impl<1>(kind, p, nwg, st);
}
int launch_stem_pool_bf16(int kind, const ContractParams& p, int nwg, hipStream_t st) {
  return launch_stem_pool_impl(kind, p, nwg, st);
}
int launch_presample_batch_bf16(const PresampleBatch& b, hipStream_t st) {
  hipLaunchKernelGGL((presample_batch_kernel<1>), dim3(b.total_blocks), dim3(256), 0, st, b);
  return (int)hipGetLastError();
}
}  // namespace btx
