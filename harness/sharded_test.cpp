// sharded_test.cpp — the multi-GPU sort from a C++ host: one process, one host thread per GPU,
// radx::ShardedSort (include/radx/radx_sharded.hpp) over an RCCL clique made by ncclCommInitAll.
//
//   sharded_test [--gpus G] [--log2n L] [--iters K] [--seed S] [--force-exchange]
//
// Every GPU generates 2^L hashed uint32 keys (the reference harness's hash, src/test/sort.cpp:226-234,
// applied to i ^ seed), the G GPUs sort them as one array, and the result is checked: every shard
// sorted, shard r holds only keys whose top log2(G) bits are r, the multiset is preserved (sum,
// xor, mixed sum over all GPUs), and for L <= 24 the concatenation of the shards equals std::sort
// of all inputs.  Prints one JSON line.  With one GPU, --force-exchange still runs the split, the
// 16 exchanges (RCCL self sends) and the per-region passes.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "radx/radx.hpp"
#include "radx/radx_sharded.hpp"

namespace {

// a reusable barrier for the host threads; abort() releases every waiter, now and later
class Barrier {
public:
    explicit Barrier(int n) : n_(n) {}
    // false when the barrier was aborted: the caller's peers will not all arrive
    bool wait() {
        std::unique_lock<std::mutex> lock(m_);
        if (aborted_) return false;
        const int gen = gen_;
        if (++count_ == n_) { count_ = 0; ++gen_; cv_.notify_all(); }
        else cv_.wait(lock, [&] { return gen != gen_ || aborted_; });
        return !aborted_;
    }
    void abort() {
        std::lock_guard<std::mutex> lock(m_);
        aborted_ = true;
        cv_.notify_all();
    }
private:
    std::mutex m_;
    std::condition_variable cv_;
    int n_, count_ = 0, gen_ = 0;
    bool aborted_ = false;
};

struct Shared {
    int gpus = 1, log2n = 22, iters = 3;
    uint32_t seed = 1;
    bool forceExchange = false;
    std::vector<std::vector<uint32_t>> inputs, shards;   // host copies for the full check
    std::vector<uint64_t> sumBefore, mixBefore, sumAfter, mixAfter, shardLen, violations, strangers;
    std::vector<double> ms;
    std::vector<std::string> errors;
    std::vector<ncclComm_t>* comms = nullptr;
    std::atomic<bool> aborted{false};
};

struct PeerFailed : std::runtime_error {
    PeerFailed() : std::runtime_error("stopped: another rank failed") {}
};

void hipOk(hipError_t e, const char* what) { radx::check_hip(e, what); }

void worker(int rank, ncclComm_t comm, Shared* sh, Barrier* barrier) {
    try {
        const int G = sh->gpus;
        const size_t n = size_t(1) << sh->log2n;
        const size_t cap = n + n / 16 + 4096;
        hipOk(hipSetDevice(rank), "hipSetDevice");
        hipStream_t stream;
        hipOk(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
        uint32_t *keys, *work, *stage;
        hipOk(hipMalloc(reinterpret_cast<void**>(&keys), cap * 4), "hipMalloc");
        hipOk(hipMalloc(reinterpret_cast<void**>(&work), cap * 4), "hipMalloc");
        hipOk(hipMalloc(reinterpret_cast<void**>(&stage), cap * 4), "hipMalloc");
        const bool keepHost = sh->log2n <= 24;
        int b = 0;
        while ((1 << b) < G) ++b;

        if (G == 1 && !sh->forceExchange) {
            // one GPU, no exchange: the plain sort through the same C ABI
            radx_hip_sorter* s = nullptr;
            radx::check_abi(radx_hip_create(rank, cap, 4, 0, &s));
            for (int it = 0; it < sh->iters + 1; ++it) {
                radx::check_abi(radx_hip_generate_u32(keys, n, sh->seed + it, stream));
                hipOk(hipStreamSynchronize(stream), "sync");
                const auto t0 = std::chrono::steady_clock::now();
                radx::check_abi(radx_hip_sort(s, keys, work, nullptr, nullptr, n, stream));
                hipOk(hipStreamSynchronize(stream), "sync");
                if (it) sh->ms[rank] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
            sh->shardLen[rank] = n;
            radx::check_abi(radx_hip_check_sorted(keys, nullptr, n, 4, &sh->violations[rank], stream));
            radx_hip_destroy(s);
        } else {
            radx::ShardedSort sorter(rank, rank, G, comm, cap, 4);
            size_t shard = 0;
            for (int it = 0; it < sh->iters + 1; ++it) {
                const uint32_t seed = (sh->seed + it) * 7919u + rank * 104729u;
                radx::check_abi(radx_hip_generate_u32(keys, n, seed, stream));
                if (it == sh->iters) {  // the last sort is the one that is checked
                    uint64_t c[3];
                    radx::check_abi(radx_hip_checksum(keys, n, 4, c, stream));
                    sh->sumBefore[rank] = c[0]; sh->mixBefore[rank] = c[2];
                    if (keepHost) {
                        sh->inputs[rank].resize(n);
                        hipOk(hipMemcpy(sh->inputs[rank].data(), keys, n * 4, hipMemcpyDeviceToHost), "hipMemcpy");
                    }
                }
                hipOk(hipStreamSynchronize(stream), "sync");
                if (!barrier->wait()) throw PeerFailed();
                const auto t0 = std::chrono::steady_clock::now();
                shard = sorter.sort(keys, work, stage, n, stream);
                hipOk(hipStreamSynchronize(stream), "sync");
                if (!barrier->wait()) throw PeerFailed();
                if (it) sh->ms[rank] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
            sh->shardLen[rank] = shard;
            uint64_t c[3];
            radx::check_abi(radx_hip_checksum(keys, shard, 4, c, stream));
            sh->sumAfter[rank] = c[0]; sh->mixAfter[rank] = c[2];
            radx::check_abi(radx_hip_check_sorted(keys, nullptr, shard, 4, &sh->violations[rank], stream));
            // sorted, so the first and last key bound every key's top bits
            if (shard && b) {
                uint32_t ends[2];
                hipOk(hipMemcpy(&ends[0], keys, 4, hipMemcpyDeviceToHost), "hipMemcpy");
                hipOk(hipMemcpy(&ends[1], keys + shard - 1, 4, hipMemcpyDeviceToHost), "hipMemcpy");
                sh->strangers[rank] = ((ends[0] >> (32 - b)) != uint32_t(rank)) + ((ends[1] >> (32 - b)) != uint32_t(rank));
            }
            if (keepHost) {
                sh->shards[rank].resize(shard);
                hipOk(hipMemcpy(sh->shards[rank].data(), keys, shard * 4, hipMemcpyDeviceToHost), "hipMemcpy");
            }
        }
        hipOk(hipFree(keys), "hipFree"); hipOk(hipFree(work), "hipFree"); hipOk(hipFree(stage), "hipFree");
        hipOk(hipStreamDestroy(stream), "hipStreamDestroy");
    } catch (const PeerFailed&) {
        // unwound by the first failing rank; its message is the one worth printing
    } catch (const std::exception& e) {
        sh->errors[rank] = e.what();
        // The other ranks are, or soon will be, blocked in the barrier or inside an RCCL call
        // waiting for this one: release the barrier and abort every communicator so that all
        // threads unwind and main() can report the error instead of hanging.
        if (!sh->aborted.exchange(true)) {
            barrier->abort();
            for (ncclComm_t c : *sh->comms) (void)ncclCommAbort(c);
        }
    }
}

}  // namespace

int main(int argc, char** argv) {
    Shared sh;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? argv[++i] : "0"; };
        if (a == "--gpus") sh.gpus = atoi(next());
        else if (a == "--log2n") sh.log2n = atoi(next());
        else if (a == "--iters") sh.iters = atoi(next());
        else if (a == "--seed") sh.seed = static_cast<uint32_t>(atoll(next()));
        else if (a == "--force-exchange") sh.forceExchange = true;
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < 1) { fprintf(stderr, "no HIP device\n"); return 1; }
    if (sh.gpus < 1 || sh.gpus > have || (sh.gpus & (sh.gpus - 1))) {
        fprintf(stderr, "--gpus must be a power of two <= %d\n", have);
        return 2;
    }
    const int G = sh.gpus;
    for (auto* v : {&sh.sumBefore, &sh.mixBefore, &sh.sumAfter, &sh.mixAfter, &sh.shardLen, &sh.violations, &sh.strangers})
        v->assign(G, 0);
    sh.ms.assign(G, 0.0);
    sh.errors.assign(G, "");
    sh.inputs.resize(G);
    sh.shards.resize(G);

    std::vector<ncclComm_t> comms(G);
    std::vector<int> devs(G);
    for (int r = 0; r < G; ++r) devs[r] = r;
    if (ncclCommInitAll(comms.data(), G, devs.data()) != ncclSuccess) { fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }

    sh.comms = &comms;
    Barrier barrier(G);
    std::vector<std::thread> threads;
    for (int r = 0; r < G; ++r) threads.emplace_back(worker, r, comms[r], &sh, &barrier);
    for (auto& t : threads) t.join();
    if (!sh.aborted) for (int r = 0; r < G; ++r) ncclCommDestroy(comms[r]);  // aborted communicators are already gone

    bool pass = true;
    for (int r = 0; r < G; ++r)
        if (!sh.errors[r].empty()) { fprintf(stderr, "rank %d: %s\n", r, sh.errors[r].c_str()); pass = false; }
    const size_t n = size_t(1) << sh.log2n;
    uint64_t total = 0, sb = 0, mb = 0, sa = 0, ma = 0;
    double ms = 0;
    for (int r = 0; r < G; ++r) {
        total += sh.shardLen[r];
        sb += sh.sumBefore[r]; mb += sh.mixBefore[r]; sa += sh.sumAfter[r]; ma += sh.mixAfter[r];
        pass = pass && sh.violations[r] == 0 && sh.strangers[r] == 0;
        ms = std::max(ms, sh.ms[r] / std::max(1, sh.iters));
    }
    const bool exchanged = !(G == 1 && !sh.forceExchange);
    pass = pass && total == n * G && (!exchanged || (sb == sa && mb == ma));
    bool fullCheck = false;
    if (pass && exchanged && sh.log2n <= 24) {
        std::vector<uint32_t> whole, sorted;
        for (int r = 0; r < G; ++r) {
            whole.insert(whole.end(), sh.shards[r].begin(), sh.shards[r].end());
            sorted.insert(sorted.end(), sh.inputs[r].begin(), sh.inputs[r].end());
        }
        std::sort(sorted.begin(), sorted.end());
        fullCheck = true;
        pass = whole == sorted;
    }
    printf("Sorting Finished: %s\n", pass ? "PASS" : "FAIL");
    printf("{\"gpus\": %d, \"keys_per_gpu\": %zu, \"iters\": %d, \"exchange\": %s, \"ms_per_sort\": %.4f, "
           "\"gkeys_per_s\": %.3f, \"full_equality_checked\": %s, \"pass\": %s}\n",
           G, n, sh.iters, exchanged ? "true" : "false", ms, ms > 0 ? n * G / ms / 1e6 : 0.0,
           fullCheck ? "true" : "false", pass ? "true" : "false");
    return pass ? 0 : 1;
}
