// One-shot all-reduce of a SMALL fp64 buffer through peer-mapped mailboxes (OPT-IN: FACL_ONESHOT_SYNCBN=1).
//
// The data-parallel step holds 14 SyncBN reductions of 0.1-16 KB (facl_amd/dist.py).  Through RCCL each is its own
// latency-bound collective, and in the default launch path each cuts the captured graph.  Here every rank owns a
// MAILBOX -- device memory exported with hipIpcGetMemHandle and opened by every peer -- with one slot per sender and call
// parity.  A call is ONE kernel launch per rank, capturable inside a graph segment:
//   1. store my n doubles into slot [parity][my rank] of EVERY rank's mailbox (over xGMI for a peer), fence at system scope,
//      then publish the call's sequence number in the slot's flag;
//   2. wait until the R flags of MY mailbox show this sequence number;
//   3. add the R buffers in RANK ORDER (every rank computes bit-identical sums) into the output.
// The sequence number lives on the device (one counter per mailbox, advanced by the kernel): a replayed graph therefore posts
// fresh numbers every replay.  Two slot sets alternate by parity: a rank can only enter call s + 2 after it has seen every peer's
// flag of call s + 1, which a peer posts only after it has finished reading call s -- so a slot is never overwritten while
// someone still reads it.
// The wait cannot hang a GPU: it gives up after FACL_MAILBOX_TIMEOUT_TICKS of the 100 MHz real-time counter (2 s), raises the
// error word and poisons the output with NaN.
// Status: rehearsed with 2 and 4 processes sharing ONE GPU (every process opens the others' handles exactly as peers would);
// what that cannot show is cross-DEVICE visibility of the stores (fine-grained memory + system-scope fences are what the
// programming model asks for; whether an xGMI peer sees them in the order written is untested here).  Never the default.
#include "common.h"
#include <string.h>

namespace {

constexpr unsigned long long MB_TIMEOUT_TICKS = 200000000ull;      // 2 s of s_memrealtime (100 MHz)
constexpr int MB_HDR = 16;                                         // doubles in front of a slot's payload: [0] = flag (as uint64), rest pad (128 B)

struct MbArgs {
    const double* in; double* out; int n; int n_max;
    char* const* boxes;                 // device array of R mailbox base pointers (index = rank; [rank] = my own)
    int rank, R;
    unsigned long long* seq;            // device counter of THIS rank's mailbox object
    unsigned* err;                      // device error word (or null)
};

__device__ __forceinline__ double* mb_slot(char* box, int parity, int sender, int R, int n_max) {
    return reinterpret_cast<double*>(box) + ((size_t)parity * R + sender) * (size_t)(MB_HDR + n_max);
}

__global__ __launch_bounds__(256) void k_mailbox_allreduce(MbArgs a) {
    __shared__ unsigned long long s_seq;
    __shared__ int s_bad;
    if (threadIdx.x == 0) { s_seq = *a.seq + 1ull; s_bad = 0; }
    __syncthreads();
    const unsigned long long seq = s_seq;
    const int parity = (int)(seq & 1ull);
    // 1. payload into every rank's mailbox (my own included: the sum below reads one place per sender)
    for (int p = 0; p < a.R; ++p) {
        double* dst = mb_slot(a.boxes[p], parity, a.rank, a.R, a.n_max) + MB_HDR;
        for (int i = threadIdx.x; i < a.n; i += blockDim.x)
            __hip_atomic_store(dst + i, a.in[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x < a.R) {                                       // one thread per destination publishes the flag behind the payload
        unsigned long long* flag = reinterpret_cast<unsigned long long*>(mb_slot(a.boxes[threadIdx.x], parity, a.rank, a.R, a.n_max));
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 2. wait for the R senders of MY mailbox
    if (threadIdx.x < a.R) {
        const unsigned long long* flag = reinterpret_cast<const unsigned long long*>(mb_slot(a.boxes[a.rank], parity, threadIdx.x, a.R, a.n_max));
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > MB_TIMEOUT_TICKS) { s_bad = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    const bool bad = s_bad != 0;
    // 3. sum in rank order
    for (int i = threadIdx.x; i < a.n; i += blockDim.x) {
        double s = 0.0;
        for (int p = 0; p < a.R; ++p)
            s += __hip_atomic_load(mb_slot(a.boxes[a.rank], parity, p, a.R, a.n_max) + MB_HDR + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        a.out[i] = bad ? __longlong_as_double(0x7ff8000000000000ll) : s;
    }
    if (threadIdx.x == 0) {
        *a.seq = seq;
        if (bad && a.err) *a.err = 1u;
    }
}

}  // namespace

// bytes of one rank's mailbox for R ranks and payloads of at most n_max doubles
extern "C" int64_t facl_mailbox_bytes(int R, int n_max) {
    if (R < 1 || n_max < 1) return 0;
    return (int64_t)2 * R * (MB_HDR + n_max) * (int64_t)sizeof(double);
}

// Allocate this rank's mailbox (fine-grained device memory where the runtime offers it, zero-filled) and export it: `handle`
// receives the 64-byte hipIpcMemHandle_t a peer passes to facl_mailbox_open.
extern "C" int facl_mailbox_alloc(int64_t bytes, void** ptr, void* handle64) {
    if (!ptr || !handle64 || bytes < 1) return FACL_E_NULL;
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMalloc(&p, (size_t)bytes);
        if (e != hipSuccess) return (int)e;
    }
    e = hipMemset(p, 0, (size_t)bytes);
    if (e != hipSuccess) return (int)e;
    e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    memcpy(handle64, &h, 64);
    *ptr = p;
    return 0;
}

extern "C" int facl_mailbox_open(const void* handle64, void** ptr) {
    if (!handle64 || !ptr) return FACL_E_NULL;
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return (int)e;
    *ptr = p;
    return 0;
}

extern "C" int facl_mailbox_close(void* ptr) { return ptr ? (int)hipIpcCloseMemHandle(ptr) : 0; }
extern "C" int facl_mailbox_free(void* ptr) { return ptr ? (int)hipFree(ptr) : 0; }

// out[0:n] = sum over the R ranks of their in[0:n], added in rank order.  boxes: DEVICE array of R mailbox pointers (this rank's
// view of every rank's mailbox, own at [rank]); seq: device uint64 counter of this rank's mailbox object (starts at 0, every
// rank calls in the same order); err: device word raised when a peer did not post within 2 s (the output is then NaN).
extern "C" int facl_mailbox_allreduce(const double* in, double* out, int n, int n_max, void* const* boxes, int rank, int R,
                                      uint64_t* seq, uint32_t* err, void* stream) {
    if (!in || !out || !boxes || !seq) return FACL_E_NULL;
    if (n < 1 || n > n_max || R < 1 || R > 64 || rank < 0 || rank >= R) return FACL_E_SHAPE;
    MbArgs a{in, out, n, n_max, (char* const*)boxes, rank, R, (unsigned long long*)seq, err};
    hipLaunchKernelGGL(k_mailbox_allreduce, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    return facl_launch_status();
}
