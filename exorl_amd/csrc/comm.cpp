// Data-parallel collective of libexorl_hip.so: RCCL sum all-reduce over xGMI, one communicator per process (= per GPU).
// SURVEY 8b `exorl_comm_init(rank, nranks, uid)`; 8e: the exchanges of one update are the critic gradients (td3_bc.py:140-142), the
// batch-global sum |Q| behind TD3+BC's lambda (:154) and the actor gradients (:158-160). The agent enqueues them itself between its
// phases once a communicator is attached (exorl_agent_set_comm), so a data-parallel step is one host call on one stream.
#include <rccl/rccl.h>

#include "kernels.h"

struct exorl_comm {
    ncclComm_t nccl = nullptr;
    int rank = 0, nranks = 1;
};

#define EXORL_CHECK_NCCL(expr)                                                                       \
    do {                                                                                             \
        ncclResult_t _r = (expr);                                                                    \
        if (_r != ncclSuccess) {                                                                     \
            ::exorl::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

namespace exorl {
int comm_allreduce_sum(exorl_comm* c, float* buf, int64_t n, hipStream_t s) {
    EXORL_REQUIRE(c && c->nccl && buf && n > 0, "comm_allreduce: bad arguments");
    EXORL_CHECK_NCCL(ncclAllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->nccl, s));
    return 0;
}
int comm_nranks(const exorl_comm* c) { return c ? c->nranks : 1; }
}  // namespace exorl

using namespace exorl;

extern "C" {

int exorl_comm_unique_id(void* id_out) {
    EXORL_REQUIRE(id_out, "comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == EXORL_COMM_ID_BYTES, "EXORL_COMM_ID_BYTES must match ncclUniqueId");
    ncclUniqueId id;
    EXORL_CHECK_NCCL(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int exorl_comm_init(int32_t rank, int32_t nranks, const void* id, exorl_comm_t** out) {
    EXORL_REQUIRE(id && out && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad arguments (rank %d of %d)", rank, nranks);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    auto* c = new exorl_comm();
    c->rank = rank; c->nranks = nranks;
    ncclResult_t r = ncclCommInitRank(&c->nccl, nranks, uid, rank);      // on the calling thread's current HIP device
    if (r != ncclSuccess) {
        set_error("comm_init: ncclCommInitRank(rank %d of %d) -> %s", rank, nranks, ncclGetErrorString(r));
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

int exorl_comm_destroy(exorl_comm_t* c) {
    if (!c) return 0;
    if (c->nccl) (void)ncclCommDestroy(c->nccl);
    delete c;
    return 0;
}

int exorl_comm_allreduce(exorl_comm_t* c, float* buf_dev, int64_t n, void* stream) {
    return comm_allreduce_sum(c, buf_dev, n, as_stream(stream));
}

}  // extern "C"
