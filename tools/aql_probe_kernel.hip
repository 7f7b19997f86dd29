// tools/aql_probe_kernel.hip -- the one-thread kernel of tools/aql_probe.cpp (built as a code object: hipcc --genco)
#include <hip/hip_runtime.h>
typedef double v2d_t __attribute__((ext_vector_type(2)));
extern "C" __global__ void aql_probe(double2* out, unsigned long long tag) {
  v2d_t g;
  g.x = 1.0;
  g.y = __longlong_as_double((long long)tag);
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(out), "v"(g) : "memory");
}
