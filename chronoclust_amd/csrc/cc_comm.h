// cc_comm.h — the exchange step of the exact multi-GPU path: every rank holds the whole
// microcluster table, scans its own share of the table rows and all-gathers one 64-byte candidate record per
// window point; everything after that runs replicated and bit-identically on every rank.
//
// Two transports behind one call:
//   RCCL   one process per GPU; ncclAllGather over xGMI on the stream of the scan that produced the records.
//          librccl is opened with dlopen when a communicator is first asked for, so single-GPU users of the
//          library never load it (and the library has no link-time dependency on it).  One communicator per
//          group serves both HIP streams of the handle by default; with CHRONOCLUST_HIP_TWO_COMMS=1 the lookahead
//          stream gets one of its own (lane 0: main stream, lane 1: lookahead stream): operations on one
//          ncclComm_t are serialised by RCCL whatever stream they are enqueued on, which ties the lookahead scans'
//          all-gathers to the in-place ones of the validation stream.  The second communicator is created from
//          an id that travels through the first (all-gather of 128 bytes); concurrent communicators have never run
//          on more than one GPU in a build session, hence opt-in.  Nothing in the library
//          waits for a collective without a bound: host waits on a stream that may hold one poll
//          ncclCommGetAsyncError and a deadline (wait_stream); on an error or when the deadline passes both
//          communicators are aborted (ncclCommAbort) and the call returns CC_ERR_COMM.
//   LOCAL  several handles of ONE process (one host thread each) that form a group: the same all-gather done
//          with stream-ordered device copies between the handles' buffers.  This is how the sharded path is
//          verified on a machine with a single GPU (two handles on GPU 0, each scanning half of the rows).
// There is no reference counterpart: the reference is a single Python thread.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

namespace cc {

struct CommErr {
    std::string what;
};

// ---- RCCL through dlopen ---------------------------------------------------------------------------------

struct RcclApi {
    void* dl = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;

    static RcclApi& get()
    {
        static RcclApi api;
        static std::once_flag once;
        std::call_once(once, [&]() {
            // the soname first: a copy that is already mapped (e.g. the one PyTorch ships) is reused
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
            for (const char* n : names) {
                api.dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                if (api.dl) break;
            }
            if (!api.dl) return;
            api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.dl, "ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.dl, "ncclCommInitRank");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.dl, "ncclCommDestroy");
            api.CommAbort = (decltype(api.CommAbort))dlsym(api.dl, "ncclCommAbort");
            api.CommGetAsyncError = (decltype(api.CommGetAsyncError))dlsym(api.dl, "ncclCommGetAsyncError");
            api.AllGather = (decltype(api.AllGather))dlsym(api.dl, "ncclAllGather");
            api.AllReduce = (decltype(api.AllReduce))dlsym(api.dl, "ncclAllReduce");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.dl, "ncclGetErrorString");
        });
        return api;
    }
    bool ok() const
    {
        return dl && GetUniqueId && CommInitRank && CommDestroy && CommAbort && CommGetAsyncError && AllGather && AllReduce &&
               GetErrorString;
    }
};

// ---- in-process group ------------------------------------------------------------------------------------

// One rendezvous object shared by the handles of a group.  A collective is two host barriers: after the first
// every rank has published its send buffer and the event that marks it ready; after the second every rank has
// enqueued its copies, so nobody overwrites a send buffer that a peer still has to read.
struct LocalGroup {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    bool broken = false;
    std::vector<const void*> send;
    std::vector<size_t> bytes;            // what each rank brought to the current exchange (must agree)
    std::vector<hipEvent_t> ready, done;  // owned by the ranks that record them

    explicit LocalGroup(int w) : world(w), send(w, nullptr), bytes(w, 0), ready(w, nullptr), done(w, nullptr) {}

    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) throw CommErr{"in-process group was abandoned by a member"};
        const unsigned long long gen = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&]() { return generation != gen || broken; });
            if (generation == gen) throw CommErr{"in-process group was abandoned by a member"};
        }
    }
    void abandon()
    {
        std::lock_guard<std::mutex> lk(mu);
        broken = true;
        cv.notify_all();
    }
};

struct Comm {
    int rank = 0, world = 1;
    // RCCL: one communicator per stream lane (0: the handle's main stream, 1: its lookahead stream); nccl[1] may be
    // null (a group set up with one communicator serves both lanes from nccl[0])
    ncclComm_t nccl[2] = {nullptr, nullptr};
    bool broken = false;  // the group failed (error, deadline, a peer left): every further exchange fails at once
    double timeout_s = 120.0;  // bound of every host wait that may hold a collective (CHRONOCLUST_HIP_COMM_TIMEOUT_S)
    std::shared_ptr<LocalGroup> local;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;  // LOCAL: this rank's two events

    bool active() const { return world > 1 || nccl[0] != nullptr; }
    bool rccl() const { return nccl[0] != nullptr; }
    ncclComm_t lane(int l) const { return (l == 1 && nccl[1]) ? nccl[1] : nccl[0]; }

    void check(ncclResult_t r, const char* what)
    {
        if (r != ncclSuccess) {
            fail_group();
            throw CommErr{std::string(what) + ": " + RcclApi::get().GetErrorString(r)};
        }
    }

    // The group is lost for this rank: in-process peers are released (their barriers throw), RCCL communicators are
    // aborted so that kernels of pending collectives end and the streams drain.  Idempotent.
    void fail_group()
    {
        if (broken) return;
        broken = true;
        if (local) local->abandon();
        for (int l = 0; l < 2; ++l)
            if (nccl[l]) {
                (void)RcclApi::get().CommAbort(nccl[l]);
                nccl[l] = nullptr;
            }
    }

    // hipStreamSynchronize with a bound.  Without an RCCL communicator a plain synchronise (in-process groups block
    // in host barriers that abandon() releases, not in the stream).  With one: poll the stream, both communicators'
    // asynchronous error state and the deadline.
    void wait_stream(hipStream_t st)
    {
        if (broken) throw CommErr{"the group has failed earlier"};
        if (!rccl()) {
            const hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) throw CommErr{std::string("hipStreamSynchronize: ") + hipGetErrorString(e)};
            return;
        }
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        for (;;) {
            const hipError_t e = hipStreamQuery(st);
            if (e == hipSuccess) return;
            if (e != hipErrorNotReady) {
                fail_group();
                throw CommErr{std::string("hipStreamQuery: ") + hipGetErrorString(e)};
            }
            if ((++spins & 63u) == 0u) {
                for (int l = 0; l < 2; ++l) {
                    ncclResult_t ar = ncclSuccess;
                    if (nccl[l] && (RcclApi::get().CommGetAsyncError(nccl[l], &ar) != ncclSuccess || (ar != ncclSuccess && ar != ncclInProgress))) {
                        const std::string msg = std::string("asynchronous RCCL error: ") + RcclApi::get().GetErrorString(ar);
                        fail_group();
                        throw CommErr{msg};
                    }
                }
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (dt > timeout_s) {
                    fail_group();
                    char buf[160];
                    snprintf(buf, sizeof buf, "a collective did not complete within %.0f s (a peer is gone or stuck); communicators aborted", timeout_s);
                    throw CommErr{buf};
                }
                if (dt > 0.002) std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        }
    }

    // recv[p * bytes .. (p + 1) * bytes) = rank p's send[0 .. bytes), for every p, ordered on `st`.
    // lane: which of the handle's two streams `st` is (selects the RCCL communicator)
    void all_gather(const void* send, void* recv, size_t bytes, hipStream_t st, int lane_idx = 0)
    {
        if (broken) throw CommErr{"the group has failed earlier"};
        if (rccl()) {
            check(RcclApi::get().AllGather(send, recv, bytes, ncclInt8, lane(lane_idx), st), "ncclAllGather");
            return;
        }
        if (!local) {
            if (recv != send && hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
                throw CommErr{"copy failed"};
            return;
        }
        LocalGroup& g = *local;
        auto chk = [&](hipError_t e, const char* what) {
            if (e != hipSuccess) {
                g.abandon();
                throw CommErr{std::string(what) + ": " + hipGetErrorString(e)};
            }
        };
        chk(hipEventRecord(ev_ready, st), "hipEventRecord");
        g.send[rank] = send;
        g.bytes[rank] = bytes;
        g.ready[rank] = ev_ready;
        g.done[rank] = ev_done;
        g.barrier();
        // the ranks run the same sequence of exchanges (their policies read the same counters): one that arrives with
        // another block size is out of step, and copying its block would read past a buffer
        for (int p = 0; p < world; ++p)
            if (g.bytes[p] != bytes) {
                g.abandon();
                throw CommErr{"ranks out of step: rank " + std::to_string(p) + " exchanges " + std::to_string(g.bytes[p]) +
                              " bytes, rank " + std::to_string(rank) + " " + std::to_string(bytes)};
            }
        for (int p = 0; p < world; ++p) {
            if (p != rank) chk(hipStreamWaitEvent(st, g.ready[p], 0), "hipStreamWaitEvent");
            char* dst = (char*)recv + (size_t)p * bytes;
            if ((const void*)dst != g.send[p])  // (in place: a rank's own block is already where it belongs)
                chk(hipMemcpyAsync(dst, g.send[p], bytes, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
        }
        chk(hipEventRecord(ev_done, st), "hipEventRecord");
        g.barrier();
        for (int p = 0; p < world; ++p)
            if (p != rank) chk(hipStreamWaitEvent(st, g.done[p], 0), "hipStreamWaitEvent");
        g.barrier();  // the events may be re-recorded only after every peer has enqueued its waits on them
    }

    void destroy()
    {
        for (int l = 0; l < 2; ++l)
            if (nccl[l]) {
                // a communicator that is still healthy is destroyed (collective, drains); a failed group was aborted
                (void)RcclApi::get().CommDestroy(nccl[l]);
                nccl[l] = nullptr;
            }
        if (local) {
            local->abandon();
            local.reset();
        }
        if (ev_ready) { (void)hipEventDestroy(ev_ready); ev_ready = nullptr; }
        if (ev_done) { (void)hipEventDestroy(ev_done); ev_done = nullptr; }
        rank = 0;
        world = 1;
        broken = false;
    }
};

}  // namespace cc
