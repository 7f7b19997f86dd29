// tools/aql_probe.cpp -- developer probe (GPU box): launch -> result round trip of a one-thread kernel dispatched (a) through HIP
// (hipExtLaunchKernelGGL-like hipModuleLaunchKernel on a stream) and (b) as an AQL packet written straight into a user-mode HSA queue.
// What would the host loop of the iterated update (one launch per pass, four to six per scan) save by writing its own packets?
// Build: tools/aql_probe.sh      Run on the GPU box: tools/aql_probe tools/aql_probe.hsaco
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>
#include <immintrin.h>

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define HSACK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m = nullptr; hsa_status_string(s_, &m); printf("HSA error %s at line %d\n", m ? m : "?", __LINE__); return 1; } } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Find { hsa_agent_t gpu; bool have_gpu = false; hsa_agent_t cpu; bool have_cpu = false; hsa_amd_memory_pool_t kernarg; bool have_kernarg = false; };
static hsa_status_t on_agent(hsa_agent_t a, void* d) {
  Find* f = static_cast<Find*>(d);
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !f->have_gpu) { f->gpu = a; f->have_gpu = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !f->have_cpu) { f->cpu = a; f->have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_pool(hsa_amd_memory_pool_t p, void* d) {
  Find* f = static_cast<Find*>(d);
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t flags = 0;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !f->have_kernarg) { f->kernarg = p; f->have_kernarg = true; }
  return HSA_STATUS_SUCCESS;
}

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: aql_probe <code object>\n"); return 2; }
  HIPCK(hipSetDevice(0));
  double2* out = nullptr;
  HIPCK(hipHostMalloc((void**)&out, 64, hipHostMallocMapped));
  memset(out, 0, 64);
  volatile unsigned long long* tagp = reinterpret_cast<volatile unsigned long long*>(out) + 1;
  // ---- (a) through HIP: the module API with a pre-built argument buffer on a stream of its own ----
  hipModule_t mod;
  hipFunction_t fn;
  HIPCK(hipModuleLoad(&mod, argv[1]));
  HIPCK(hipModuleGetFunction(&fn, mod, "aql_probe"));
  hipStream_t st;
  HIPCK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  struct { double2* out; unsigned long long tag; } args;
  std::vector<double> rt_hip, cpu_hip;
  for (int i = 0; i < 2200; i++) {
    args.out = out; args.tag = 0x1000ull + i;
    size_t sz = sizeof(args);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    const double t0 = now_us();
    HIPCK(hipModuleLaunchKernel(fn, 1, 1, 1, 1, 1, 1, 0, st, nullptr, extra));
    const double t1 = now_us();
    while (*tagp != args.tag) _mm_pause();
    const double t2 = now_us();
    if (i >= 200) { rt_hip.push_back(t2 - t0); cpu_hip.push_back(t1 - t0); }
  }
  HIPCK(hipStreamSynchronize(st));
  // ---- (b) an AQL packet of our own ----
  HSACK(hsa_init());
  Find f;
  HSACK(hsa_iterate_agents(on_agent, &f));
  if (!f.have_gpu || !f.have_cpu) { printf("no GPU / CPU agent\n"); return 1; }
  HSACK(hsa_amd_agent_iterate_memory_pools(f.cpu, on_pool, &f));
  if (!f.have_kernarg) { printf("no kernarg pool\n"); return 1; }
  hsa_queue_t* q = nullptr;
  HSACK(hsa_queue_create(f.gpu, 256, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
  std::ifstream fs(argv[1], std::ios::binary);
  std::vector<char> blob((std::istreambuf_iterator<char>(fs)), std::istreambuf_iterator<char>());
  hsa_code_object_reader_t rd;
  HSACK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &rd));
  hsa_executable_t ex;
  HSACK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex));
  HSACK(hsa_executable_load_agent_code_object(ex, f.gpu, rd, nullptr, nullptr));
  HSACK(hsa_executable_freeze(ex, nullptr));
  hsa_executable_symbol_t sym;
  HSACK(hsa_executable_get_symbol_by_name(ex, "aql_probe.kd", &f.gpu, &sym));
  uint64_t kobj = 0; uint32_t karg = 0, lds = 0, priv = 0;
  HSACK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
  HSACK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &karg));
  HSACK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &lds));
  HSACK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &priv));
  char* kbuf = nullptr;                                   // a ring of argument blocks in the kernarg pool (host memory the GPU reads)
  const size_t kslot = 256, kslots = 64;
  HSACK(hsa_amd_memory_pool_allocate(f.kernarg, kslot * kslots, 0, (void**)&kbuf));
  HSACK(hsa_amd_agents_allow_access(1, &f.gpu, nullptr, kbuf));
  printf("kernel object %#llx, kernarg %u B, LDS %u, scratch %u; queue of %u packets\n", (unsigned long long)kobj, karg, lds, priv, q->size);
  std::vector<double> rt_aql, cpu_aql;
  const uint32_t mask = q->size - 1;
  for (int i = 0; i < 2200; i++) {
    const unsigned long long tag = 0x200000ull + i;
    const double t0 = now_us();
    char* ka = kbuf + (size_t)(i % kslots) * kslot;
    memset(ka, 0, karg);
    memcpy(ka, &out, 8);
    memcpy(ka + 8, &tag, 8);
    const uint64_t wi = hsa_queue_add_write_index_relaxed(q, 1);
    hsa_kernel_dispatch_packet_t* p = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(q->base_address) + (wi & mask);
    p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    p->workgroup_size_x = 1; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
    p->grid_size_x = 1; p->grid_size_y = 1; p->grid_size_z = 1;
    p->private_segment_size = priv; p->group_segment_size = lds;
    p->kernel_object = kobj;
    p->kernarg_address = ka;
    p->completion_signal.handle = 0;
    const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                            (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                            (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    __atomic_store_n(reinterpret_cast<uint16_t*>(p), header, __ATOMIC_RELEASE);
    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)wi);
    const double t1 = now_us();
    while (*tagp != tag) _mm_pause();
    const double t2 = now_us();
    if (i >= 200) { rt_aql.push_back(t2 - t0); cpu_aql.push_back(t1 - t0); }
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto q1 = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 4]; };
  printf("HIP  (hipModuleLaunchKernel, argument buffer): round trip median %.2f us (lower quartile %.2f), of it the call %.2f us\n", med(rt_hip), q1(rt_hip), med(cpu_hip));
  printf("AQL  (own packet, own queue)                 : round trip median %.2f us (lower quartile %.2f), of it the writes %.2f us\n", med(rt_aql), q1(rt_aql), med(cpu_aql));
  hsa_queue_destroy(q);
  return 0;
}
