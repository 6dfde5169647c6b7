// fp32 MFMA kernel for the L1 tensor product (placeholder until the kernel lands).
#include "e3_common.h"
namespace e3 {
int mfma_plan_init(e3_l1tp_plan*) { return E3_OK; }
int mfma_plan_upload(e3_l1tp_plan*) { return E3_OK; }
void mfma_plan_free(e3_l1tp_plan*) {}
bool mfma_supported(const e3_l1tp_plan*, int) { return false; }
int64_t mfma_packed_bytes(const e3_l1tp_plan*) { return 0; }
int mfma_pack(const e3_l1tp_plan*, const void* const[4], const void* const[4], int, void*, hipStream_t) { return E3_OK; }
int mfma_forward(const e3_l1tp_plan*, const void*, int64_t, const void*, int64_t, const void*, void*, int64_t, int64_t, int, hipStream_t) { return E3_ERR_UNSUPPORTED; }
}
