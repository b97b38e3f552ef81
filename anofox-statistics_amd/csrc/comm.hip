// comm.hip — the multi-GPU exchange step behind the C ABI: one RCCL all-gather of the per-group
// records over xGMI, callable from a DuckDB extension (or any host) without Python.
//
// The path shards by GROUP BY key: every rank fits its own groups with no data-path communication, then every rank
// needs every group's record — one collective of fixed-size f64 records (p + 6 doubles per group, + 5p + 2 with
// inference).  There is no reference counterpart (the reference is single-process CPU code).
//
// RCCL is loaded at run time (dlopen of librccl.so.1, local scope) by the first anofox_hip_comm_* call, so the
// library itself carries no link-time dependency on it: processes that never gather across GPUs (and the CPU-only
// build container) do not need RCCL at all, and a process that already has an RCCL loaded (PyTorch ships one) keeps
// using that copy.
#include <dlfcn.h>
#include <stdlib.h>
#include <rccl/rccl.h>

#include "context.h"

using namespace anofox;
using namespace anofox::host;

struct AnofoxHipComm {
	AnofoxHipContext *ctx = nullptr;
	ncclComm_t comm = nullptr;
	int world = 1, rank = 0;
};

namespace {

struct Rccl {
	void *handle = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
	std::string error;
};

Rccl &rccl() {
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, [] {
		// ANOFOX_RCCL_LIB: the one library to load (deployments with RCCL elsewhere; the test of the error path below)
		const char *only = getenv("ANOFOX_RCCL_LIB");
		const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
		if (only && *only) {
			r.handle = dlopen(only, RTLD_NOW | RTLD_LOCAL);
		} else {
			for (const char *n : names) {
				r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
				if (r.handle) break;
			}
		}
		if (!r.handle) {
			const char *msg = dlerror(); // (reading it clears it: once)
			r.error = std::string("RCCL is not available: ") + (msg ? msg : "dlopen failed");
			return;
		}
		r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
		r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
		r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
		r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
		r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
		r.CommCount = (decltype(r.CommCount))dlsym(r.handle, "ncclCommCount");
		if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) r.error = "RCCL symbols missing in librccl";
	});
	return r;
}

bool rccl_fail(ncclResult_t rc, const char *what, AnofoxError *e) {
	if (rc == ncclSuccess) return false;
	Rccl &r = rccl();
	set_error(e, ANOFOX_ERROR_INTERNAL, std::string("RCCL error in ") + what + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "?"));
	return true;
}

} // namespace

extern "C" {

bool anofox_hip_comm_unique_id(uint8_t *out_id, AnofoxError *out_error) {
	reset_error(out_error);
	if (!out_id) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "out_id is NULL"); return false; }
	Rccl &r = rccl();
	if (!r.error.empty()) { set_error(out_error, ANOFOX_ERROR_INTERNAL, r.error); return false; }
	ncclUniqueId id;
	if (rccl_fail(r.GetUniqueId(&id), "ncclGetUniqueId", out_error)) return false;
	static_assert(sizeof id == ANOFOX_HIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
	memcpy(out_id, &id, sizeof id);
	return true;
}

bool anofox_hip_comm_create(AnofoxHipContext *ctx, int world_size, int rank, const uint8_t *id, AnofoxHipComm **out_comm,
                            AnofoxError *out_error) {
	reset_error(out_error);
	if (!out_comm) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "out_comm is NULL"); return false; }
	*out_comm = nullptr;
	if (!ctx || !id || world_size < 1 || rank < 0 || rank >= world_size) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "context or id is NULL, or rank / world_size out of range");
		return false;
	}
	Rccl &r = rccl();
	if (!r.error.empty()) { set_error(out_error, ANOFOX_ERROR_INTERNAL, r.error); return false; }
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	ncclUniqueId uid;
	memcpy(&uid, id, sizeof uid);
	auto *c = new (std::nothrow) AnofoxHipComm();
	if (!c) { set_error(out_error, ANOFOX_ERROR_ALLOCATION_FAILURE, "communicator allocation failed"); return false; }
	c->ctx = ctx;
	c->world = world_size;
	c->rank = rank;
	if (rccl_fail(r.CommInitRank(&c->comm, world_size, uid, rank), "ncclCommInitRank", out_error)) {
		delete c;
		return false;
	}
	*out_comm = c;
	return true;
}

void anofox_hip_comm_destroy(AnofoxHipComm *comm) {
	if (!comm) return;
	if (comm->comm) {
		if (comm->ctx) {
			(void)hipSetDevice(comm->ctx->device);
			(void)hipStreamSynchronize(comm->ctx->stream);
		}
		(void)rccl().CommDestroy(comm->comm);
	}
	delete comm;
}

int anofox_hip_comm_world_size(const AnofoxHipComm *comm) { return comm ? comm->world : 0; }
// ranks RCCL itself counts in the communicator (ncclCommCount) — what an N-GPU run can be audited with; 0 if unknown
int anofox_hip_comm_ranks_seen(const AnofoxHipComm *comm) {
	if (!comm || !comm->comm) return 0;
	Rccl &r = rccl();
	int n = 0;
	if (!r.CommCount || r.CommCount(comm->comm, &n) != ncclSuccess) return 0;
	return n;
}
int anofox_hip_comm_rank(const AnofoxHipComm *comm) { return comm ? comm->rank : -1; }

bool anofox_hip_gather_records_device(AnofoxHipComm *comm, const double *d_local, int64_t records_per_rank, size_t record_len,
                                      double *d_all, AnofoxError *out_error) {
	reset_error(out_error);
	if (!comm || records_per_rank < 0 || record_len == 0) {
		set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "communicator is NULL, or negative count / zero record length");
		return false;
	}
	if (records_per_rank == 0) return true;
	if (!d_local || !d_all) { set_error(out_error, ANOFOX_ERROR_INVALID_INPUT, "record buffers are NULL"); return false; }
	AnofoxHipContext *ctx = comm->ctx;
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hip_fail(hipSetDevice(ctx->device), "hipSetDevice", out_error)) return false;
	// stream-ordered behind the fit kernels of the same context: no host synchronisation between fit and gather
	return !rccl_fail(rccl().AllGather(d_local, d_all, (size_t)records_per_rank * record_len, ncclDouble, comm->comm, ctx->stream),
	                  "ncclAllGather", out_error);
}

} // extern "C"
