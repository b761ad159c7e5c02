// TEST INFRASTRUCTURE ONLY (never loaded by the product unless NWKNN_RCCL_LIBRARY points at it): a stand-in for librccl.so whose "ranks" may all
// live on ONE GPU.  It lets a one-GPU box execute the N > 1 path of nwknn_search_sharded end to end -- a host thread per virtual device, shard
// bounds and index offsets, the all-gather's buffer layout, every device's merge of ITS slice of the query rows at the lists' own stride, the
// per-slice copies to the caller -- with everything real except the transport: ncclAllGather here is N device-to-device copies per rank on the
// rank's stream.  The marker symbol tells nwknn_comm_create that this transport accepts a device listed more than once (real RCCL refuses).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

struct FakeComm { int rank, n; };
struct Call { const void* send; void* recv; size_t bytes; int rank, n; hipStream_t stream; };

static std::mutex g_mu;
static std::vector<Call> g_pending;
static int g_depth = 0;
static long long g_allgathers = 0;
static long long g_fail_at = -1;                 // nwknn_test_transport_fail_call(n): the n-th ncclAllGather from now fails once (error-path tests)

static size_t elem_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

static ncclResult_t flush_locked()
{
    // the j-th call of every rank belongs to collective j (nwknn_search_sharded issues [distances, indices] per device, in that order)
    int n = 0;
    for (const Call& c : g_pending) n = c.n > n ? c.n : n;
    std::vector<std::vector<Call>> per_rank((size_t)n);
    for (const Call& c : g_pending) per_rank[(size_t)c.rank].push_back(c);
    g_pending.clear();
    for (int r = 0; r < n; ++r)
        for (size_t j = 0; j < per_rank[(size_t)r].size(); ++j) {
            const Call& me = per_rank[(size_t)r][j];
            for (int s = 0; s < n; ++s) {
                if (j >= per_rank[(size_t)s].size()) return ncclInvalidUsage;       // a rank missing from the collective
                const Call& src = per_rank[(size_t)s][j];
                if (src.bytes != me.bytes) return ncclInvalidArgument;
                if (hipMemcpyAsync((char*)me.recv + (size_t)s * me.bytes, src.send, me.bytes, hipMemcpyDeviceToDevice, me.stream) != hipSuccess)
                    return ncclUnhandledCudaError;
            }
        }
    return ncclSuccess;
}

extern "C" {

int nwknn_test_transport_shares_one_device = 1;

ncclResult_t ncclGetVersion(int* v) { if (v) *v = 0; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake RCCL error"; }

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* /*devlist*/)
{
    for (int i = 0; i < ndev; ++i) comms[i] = (ncclComm_t) new FakeComm{i, ndev};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) { delete (FakeComm*)c; return ncclSuccess; }

ncclResult_t ncclGroupStart() { std::lock_guard<std::mutex> l(g_mu); ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
    std::lock_guard<std::mutex> l(g_mu);
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    const ncclResult_t r = flush_locked();
    g_pending.clear();
    return r;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream)
{
    std::lock_guard<std::mutex> l(g_mu);
    const FakeComm* c = (const FakeComm*)comm;
    if (g_fail_at > 0 && --g_fail_at == 0) { g_fail_at = -1; return ncclInternalError; }
    g_pending.push_back(Call{send, recv, count * elem_bytes(dt), c->rank, c->n, stream});
    ++g_allgathers;
    if (g_depth == 0) return c->n == 1 ? flush_locked() : ncclInvalidUsage;       // several ranks in one process must be grouped
    return ncclSuccess;
}

long long nwknn_test_transport_allgather_calls() { return g_allgathers; }
void nwknn_test_transport_fail_call(long long n) { std::lock_guard<std::mutex> l(g_mu); g_fail_at = n; }
int nwknn_test_transport_group_depth() { std::lock_guard<std::mutex> l(g_mu); return g_depth; }
long long nwknn_test_transport_pending() { std::lock_guard<std::mutex> l(g_mu); return (long long)g_pending.size(); }

}  // extern "C"
