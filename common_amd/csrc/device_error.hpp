// device_error.hpp -- how a kernel tells the host that something it relies on did not hold (a wave-subset barrier that
// timed out, an entity leaving a group it is not in, a relation offset past the score row).  One pinned, host-visible
// pair of words per device {code bits, detail}; every kernel translation unit keeps its own copy of the pointer (the
// library is built without relocatable device code) and the context binds them all when it is created.  The host reads
// the word at its synchronisation points and at the entry of the launching calls (abi.cpp device_error_check): the
// failing call returns MSC_EDEVICE and the tables of the state it touched are to be rebuilt (msc_accumulate with
// MSC_ACC_RESET).
#pragma once
#include "family_math.hpp"

namespace msc {

enum : uint32_t {
  MSC_DEVERR_BARRIER_TIMEOUT = 1u,   // WaveSubsetBarrier gave up waiting (score_block.hpp); detail = blockIdx.x
  MSC_DEVERR_ENTITY_OP = 2u,         // k_entity_op: leave from an empty group / a group the row is not in, join of an assigned row; detail = group
  MSC_DEVERR_RELATION_RANGE = 4u,    // k_relation_slice_scores: off[c] + g * stride beyond the score row; detail = cell
};

static __device__ uint32_t *g_dev_error = nullptr;

MSC_DEV void report_device_error(uint32_t code, uint32_t detail) {
  uint32_t *w = g_dev_error;
  if (w == nullptr) return;
  // code BITS: a second error adds its bit, the detail stays the first one's (the host reads and clears both)
  const uint32_t before = __hip_atomic_fetch_or(w, code, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM);
  if (before == 0u) __hip_atomic_store(w + 1, detail, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the body of every translation unit's bind_error_word_*(): point this unit's copy at the device's word
#define MSC_DEFINE_BIND_ERROR_WORD(name)                                                             \
  int name(uint32_t *word_dev) {                                                                     \
    return hipMemcpyToSymbol(HIP_SYMBOL(g_dev_error), &word_dev, sizeof(word_dev)) == hipSuccess ? 0 : -1; \
  }

}  // namespace msc
