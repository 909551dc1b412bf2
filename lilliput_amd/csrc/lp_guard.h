// lp_guard.h -- every device / pinned allocation of the library goes through here.
//
// Normally these are hipMalloc / hipFree and hipHostMalloc / hipHostFree. With LILLIPUT_HIP_GUARD=<alignment> (1 = 64 bytes) they become
// guard-page allocations, the device-side equivalent of an electric fence, so that a kernel or a DMA transfer that touches one byte
// outside a buffer faults on the spot instead of reading whatever happens to be mapped next to it (the reference never reads outside
// the caller's buffer: /root/reference/opencv.cpp:99-124 wraps the Go slice in a cv::Mat of exactly its length):
//   device  : hipMemAddressReserve of the buffer + one granule on either side, hipMemCreate + hipMemMap of the middle only -- the granule
//             before and the one after stay UNMAPPED -- and the buffer placed flush against the END of the mapping (rounded up to the
//             alignment asked for). The gap between the start of the mapping and the buffer carries a canary pattern that is checked
//             when the buffer is freed (an under-run that stays inside the mapping).
//   pinned  : an anonymous mmap with a PROT_NONE page on either side, hipHostRegister'ed (mapped, portable) in between, buffer flush
//             against the end; the page after it is neither CPU- nor GPU-accessible.
// A guarded buffer is allocated at exactly the size asked for (no geometric growth), so the arenas re-allocate more often: a
// debugging mode, several times slower on mixed batches. LILLIPUT_HIP_GUARD_LOG=1 prints one line per allocation (tag, size, address
// range) -- the "Memory access fault ... on address X" line of the runtime then names the buffer whose end X is.
// scripts/archive/r04_guard.sh runs the GPU test suite, smoke() and every bench workload under it (profiles/r04_guard.md).
#pragma once
#include <stddef.h>

bool lp_guard_on();
// 0 on success. tag: a static string naming the arena (log lines only).
int lp_dev_malloc(void** p, size_t bytes, const char* tag);
void lp_dev_free(void* p);
// pinned host memory, device-mapped (hipHostMallocMapped, + hipHostMallocPortable when portable)
int lp_pinned_malloc(void** p, size_t bytes, bool portable, const char* tag);
void lp_pinned_free(void* p);
