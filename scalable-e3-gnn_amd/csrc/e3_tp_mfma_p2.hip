// Instantiations of tp_fwd_mfma_kernel, part 2 (see e3_tp_mfma_kernel.h): l_max 2 message TP #2 and update TP #1
#include "e3_common.h"
#include "cg_tables.h"
#include "e3_tp_internal.h"

#include <type_traits>
#include <vector>

namespace e3 {

#include "e3_tp_mfma_core.h"
#include "e3_tp_mfma_kernel.h"

std::vector<FastKernelEntry> fast_kernels_part2() {
  return {
      E3_FAST(2, 3, 1, 1, 0, 1, 2), E3_FAST(2, 3, 1, 1, 0, 1, 2, 0, 1, 2),
  };
}

}  // namespace e3
