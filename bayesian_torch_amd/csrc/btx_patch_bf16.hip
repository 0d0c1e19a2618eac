// Instantiates the bf16 "patch" variant (stride-1 2-D convolutions, halo'd input patch resident in LDS).
#include "btx_contract_patch.h"
namespace btx {
int launch_contract_patch_bf16(int kind, const ContractParams& p, int nwg, hipStream_t st) {
  return launch_contract_patch_impl<1>(kind, p, nwg, st);
}
}  // namespace btx
