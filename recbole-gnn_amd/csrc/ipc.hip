// ipc.hip — peer-to-peer halo PUSH without a collective (r06; SURVEY.md §8(e): "hipMemcpyPeerAsync ring / direct over xGMI").
//
// The node-range sharded layer (sharded.py / shard.hip) exchanges halo rows per layer.  With a collective that is three passes —
// pack into a send buffer, all-to-all, (the receiver gathers from its receive buffer) — and one RCCL launch per layer.  Here a
// rank's layer tables [owned rows | halo rows] live in memory it EXPORTS (hipIpcGetMemHandle); every peer maps them
// (hipIpcOpenMemHandle: the same HBM when both processes share a GPU, a peer mapping over xGMI otherwise) and its pack kernel
// (rbg_gather_rows_f32) stores the rows the owner needs STRAIGHT into the owner's table tail.  Ordering is by two small flag
// kernels on the ranks' own streams:
//   rbg_ipc_signal : after the pushes of a layer, release-store a sequence number into the receiver's flag word (one per sender
//                    and table);
//   rbg_ipc_wait   : the receiver's stream spins (bounded: a time-out sets an error word instead of hanging the GPU) until all
//                    its senders' words have reached the sequence number, then its layer launch follows in stream order.
// Nothing here is a collective and nothing touches the host after set-up.  The reference has no multi-GPU path at all (no
// interface to mirror); what is exchanged are the rows of LightGCNConv's operand (layers.py:13-20).
//
// Real peers (different GPUs) could not be exercised on the one-GPU test box: two processes on ONE device run it functionally
// (tests/test_gpu_sharded.py); sharded.py keeps the collective as the default transport.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "internal.h"

namespace rbg {

__global__ void ipc_signal_kernel(unsigned long long *flag, unsigned long long value) {
    __threadfence_system();  // (the stream's earlier kernels — the pushes — have completed: their stores are released with the flag)
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one thread per flag word; spins until the word has reached `value` or ~timeout_ticks of the 100 MHz real-time counter pass
__global__ void ipc_wait_kernel(const unsigned long long *flags, int n, unsigned long long value, unsigned long long timeout_ticks, int *err) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(flags + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
        __builtin_amdgcn_s_sleep(32);
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
            atomicExch(err, 1 + i);
            return;
        }
    }
    __threadfence_system();
}

}  // namespace rbg

using namespace rbg;

extern "C" {

int rbg_ipc_alloc(void **ptr, int64_t bytes, int device) {
    clear_error();
    if (!ptr || bytes <= 0 || device < 0) return fail(RBG_EINVAL, "bad argument");
    *ptr = nullptr;
    int rc = set_device_for(device);
    if (rc) return rc;
    if (hipMalloc(ptr, (size_t)bytes) != hipSuccess) {
        (void)hipGetLastError();
        return fail(RBG_ENOMEM, "device allocation of %lld bytes failed", (long long)bytes);
    }
    if (hipMemset(*ptr, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(*ptr);
        *ptr = nullptr;
        return fail(RBG_EHIP, "clearing the allocation failed");
    }
    return RBG_OK;
}

void rbg_ipc_free(void *ptr) {
    if (ptr) (void)hipFree(ptr);
}

int rbg_ipc_export(void *ptr, void *handle) {
    clear_error();
    if (!ptr || !handle) return fail(RBG_EINVAL, "NULL argument");
    static_assert(sizeof(hipIpcMemHandle_t) == RBG_IPC_HANDLE_BYTES, "RBG_IPC_HANDLE_BYTES must equal sizeof(hipIpcMemHandle_t)");
    RBG_HIP(hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t *>(handle), ptr));
    return RBG_OK;
}

int rbg_ipc_open(const void *handle, void **ptr, int device) {
    clear_error();
    if (!ptr || !handle || device < 0) return fail(RBG_EINVAL, "bad argument");
    *ptr = nullptr;
    int rc = set_device_for(device);
    if (rc) return rc;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof h);
    RBG_HIP(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
    return RBG_OK;
}

int rbg_ipc_close(void *ptr) {
    clear_error();
    if (!ptr) return RBG_OK;
    RBG_HIP(hipIpcCloseMemHandle(ptr));
    return RBG_OK;
}

int rbg_ipc_signal(void *flag, uint64_t value, void *stream) {
    clear_error();
    if (!flag) return fail(RBG_EINVAL, "flag is NULL");
    hipLaunchKernelGGL(ipc_signal_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long *>(flag), (unsigned long long)value);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

int rbg_ipc_wait(const void *flags, int n, uint64_t value, int timeout_ms, int *err, void *stream) {
    clear_error();
    if (!flags || !err || n < 1 || n > 64 || timeout_ms <= 0) return fail(RBG_EINVAL, "bad argument (1 <= n <= 64 flag words)");
    hipLaunchKernelGGL(ipc_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<const unsigned long long *>(flags), n,
                       (unsigned long long)value, (unsigned long long)timeout_ms * 100000ull, err);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

}  // extern "C"
