/**
 * @file multi_gpu.hip
 * @brief C ABI: process-per-GPU job attachment (RCCL) and vertex-partitioned runs.
 */
#include "capi_internal.hxx"

using namespace essentials_amd;

extern "C" {

int grx_comm_unique_id(void*) { return unsupported("multi-GPU: not built yet"); }
int grx_comm_attach(grx_context_t, const void*, int, int) { return unsupported("multi-GPU: not built yet"); }
int grx_comm_detach(grx_context_t) { return unsupported("multi-GPU: not built yet"); }
int grx_graph_rmat_partition(grx_context_t, uint32_t, uint32_t, uint64_t, uint64_t, int,
                             grx_graph_t*, int32_t*, int32_t*) {
  return unsupported("multi-GPU: not built yet");
}
int grx_graph_partition(grx_context_t, grx_graph_t, grx_graph_t*, int32_t*, int32_t*) {
  return unsupported("multi-GPU: not built yet");
}
int grx_bfs_partitioned(grx_context_t, grx_graph_t, int32_t, int32_t, int32_t, int32_t, int32_t*,
                        const grx_options*, grx_stats*) {
  return unsupported("multi-GPU: not built yet");
}
int grx_sssp_partitioned(grx_context_t, grx_graph_t, int32_t, int32_t, int32_t, int32_t, float*,
                         const grx_options*, grx_stats*) {
  return unsupported("multi-GPU: not built yet");
}

}  // extern "C"
