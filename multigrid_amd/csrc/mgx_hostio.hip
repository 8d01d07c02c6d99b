// mgx_hostio.hip -- moving a caller's ORDINARY host arrays to and from the device fast.
//
// The reference's seam hands over pageable host memory (DTMutableDoubleArray::Pointer(),
// main.cpp:445; DTSource allocates with new double[], DTDoubleArray.cpp:39).  A plain
// hipMemcpy of such memory is staged by the runtime through one pinned buffer on one
// thread: about 29 GB/s up and 17 GB/s down on the test box, which at 16385^2 (2.15 GB per
// array) costs far more than the V-cycles themselves (285 ms of transfers around 26 ms of
// cycles; with the code below 135 ms, about 50 GB/s each way).  Here several host threads each copy
// their share of the array through their own pair of pinned staging buffers and their own
// stream, so the host-side memcpy of one chunk overlaps the DMA of another and the PCIe
// link, not a single core, sets the rate.  The pinned buffers are allocated once per
// process and kept.  Page-locking the caller's own arrays (hipHostRegister) would avoid
// the staging copy altogether but is refused on some systems, the test box among them.
#include "mgx_internal.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

namespace mgx {

namespace {

constexpr int MAX_THREADS = 16;

// one staging buffer (MGX_IO_CHUNK_MB, default 4): 8 threads x 2 x 4 MB sustain the same rate
// as 16 MB buffers (measured: 16385^2, 2 x 2.15 GB up + 2.15 GB down + 10 cycles in 161 ms
// with 4, 8 or 16 MB), so the page-locked pool is kept small
size_t chunk_bytes()
{
    static const size_t n = [] {
        const char *e = getenv("MGX_IO_CHUNK_MB");
        long mb = e ? atol(e) : 4;
        if (mb < 1) mb = 1;
        if (mb > 256) mb = 256;
        return (size_t)mb << 20;
    }();
    return n;
}
#define CHUNK_BYTES (chunk_bytes())

struct Staging {
    std::mutex mu;  // one staged transfer at a time uses the pool
    int nthreads = 0;
    void *buf[MAX_THREADS][2] = {{nullptr}};
    bool tried = false, ok = false;
};

Staging &pool()
{
    static Staging s;
    return s;
}

int io_threads()
{
    const char *e = getenv("MGX_IO_THREADS");
    int n = e ? atoi(e) : 8;
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && n > hw) n = hw;
    if (n < 1) n = 1;
    if (n > MAX_THREADS) n = MAX_THREADS;
    return n;
}

// task(t) for t = 0..n-1, each on a thread of its own where one can be created; a task whose
// thread cannot be created (std::system_error must not cross the C ABI) runs on the calling
// thread after that thread's own share.  The tasks are independent of each other.
template <class F>
void run_tasks(int n, F task)
{
    std::vector<std::thread> th;
    std::vector<int> here;
    for (int t = 1; t < n; t++) {
        try {
            th.emplace_back(task, t);
        } catch (...) {
            here.push_back(t);
        }
    }
    task(0);
    for (int t : here) task(t);
    for (std::thread &x : th) x.join();
}

// pinned staging buffers, allocated on first use (false: none, use plain copies)
bool ensure_pool(Staging &P)
{
    if (P.tried) return P.ok;
    P.tried = true;
    P.nthreads = io_threads();
    // page-locking is slow and serial per call: let every thread lock its own pair
    std::vector<int> bad((size_t)P.nthreads, 0);
    run_tasks(P.nthreads, [&P, &bad](int t) {
        for (int k = 0; k < 2; k++)
            if (hipHostMalloc(&P.buf[t][k], CHUNK_BYTES, hipHostMallocPortable) != hipSuccess) {
                (void)hipGetLastError();
                P.buf[t][k] = nullptr;
                bad[(size_t)t] = 1;
            }
    });
    bool ok = true;
    for (int b : bad) ok = ok && !b;
    if (!ok) {
        for (int a = 0; a < P.nthreads; a++)
            for (int b = 0; b < 2; b++)
                if (P.buf[a][b]) {
                    (void)hipHostFree(P.buf[a][b]);
                    P.buf[a][b] = nullptr;
                }
        return false;
    }
    P.ok = true;
    return true;
}

struct Worker {
    hipError_t err = hipSuccess;
};

// thread t handles chunks t, t+T, t+2T, ...; to_device: host -> device, else device -> host
void run_worker(int t, int T, int device, bool to_device, char *dev, char *host, size_t nbytes, void *const bufs[2],
                Worker *w)
{
    hipError_t e = hipSetDevice(device);
    hipStream_t s = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int k = 0; k < 2 && e == hipSuccess; k++) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
    const size_t nchunks = (nbytes + CHUNK_BYTES - 1) / CHUNK_BYTES;
    if (e == hipSuccess) {
        if (to_device) {
            int i = 0;
            for (size_t c = (size_t)t; c < nchunks && e == hipSuccess; c += (size_t)T, i++) {
                const size_t off = c * CHUNK_BYTES, n = std::min(CHUNK_BYTES, nbytes - off);
                const int b = i & 1;
                if (i >= 2) e = hipEventSynchronize(ev[b]);  // the DMA that last read this buffer
                if (e != hipSuccess) break;
                memcpy(bufs[b], host + off, n);
                e = hipMemcpyAsync(dev + off, bufs[b], n, hipMemcpyHostToDevice, s);
                if (e == hipSuccess) e = hipEventRecord(ev[b], s);
            }
        } else {
            // keep one DMA in flight while the previous chunk is copied out to the caller's array
            std::vector<size_t> mine;
            for (size_t c = (size_t)t; c < nchunks; c += (size_t)T) mine.push_back(c);
            auto issue = [&](size_t i) {
                const size_t off = mine[i] * CHUNK_BYTES, n = std::min(CHUNK_BYTES, nbytes - off);
                hipError_t r = hipMemcpyAsync(bufs[i & 1], dev + off, n, hipMemcpyDeviceToHost, s);
                if (r == hipSuccess) r = hipEventRecord(ev[i & 1], s);
                return r;
            };
            if (!mine.empty()) e = issue(0);
            for (size_t i = 0; i < mine.size() && e == hipSuccess; i++) {
                if (i + 1 < mine.size()) e = issue(i + 1);
                if (e == hipSuccess) e = hipEventSynchronize(ev[i & 1]);
                if (e != hipSuccess) break;
                const size_t off = mine[i] * CHUNK_BYTES, n = std::min(CHUNK_BYTES, nbytes - off);
                memcpy(host + off, bufs[i & 1], n);
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    for (int k = 0; k < 2; k++)
        if (ev[k]) (void)hipEventDestroy(ev[k]);
    if (s) (void)hipStreamDestroy(s);
    w->err = e;
}

hipError_t staged(bool to_device, void *dev, void *host, size_t nbytes, hipStream_t order)
{
    Staging &P = pool();
    std::unique_lock<std::mutex> lock(P.mu, std::try_to_lock);
    static const size_t min_bytes = 64u << 20;
    if (nbytes < min_bytes || !lock.owns_lock() || !ensure_pool(P)) {
        // small, or the pool is busy or unavailable: the runtime's own path
        hipError_t e = to_device ? hipMemcpyAsync(dev, host, nbytes, hipMemcpyHostToDevice, order)
                                 : hipMemcpyAsync(host, dev, nbytes, hipMemcpyDeviceToHost, order);
        if (e == hipSuccess) e = hipStreamSynchronize(order);
        return e;
    }
    // everything enqueued on the caller's stream so far comes first (the workers use their own streams)
    hipError_t e = hipStreamSynchronize(order);
    if (e != hipSuccess) return e;
    int device = 0;
    e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    const int T = P.nthreads;
    std::vector<Worker> w((size_t)T);
    run_tasks(T, [&](int t) { run_worker(t, T, device, to_device, (char *)dev, (char *)host, nbytes, P.buf[t], &w[t]); });
    for (const Worker &x : w)
        if (x.err != hipSuccess) return x.err;
    return hipSuccess;
}

} // namespace

hipError_t host_to_device(double *dst_dev, const double *src_host, size_t count, hipStream_t order)
{
    return staged(true, dst_dev, const_cast<double *>(src_host), sizeof(double) * count, order);
}

hipError_t device_to_host(double *dst_host, const double *src_dev, size_t count, hipStream_t order)
{
    return staged(false, const_cast<double *>(src_dev), dst_host, sizeof(double) * count, order);
}

// Whether every interior entry of an m x m column-major host array has all bits zero (+0.0;
// a -0.0 counts as data).  Several threads, each over a block of columns; all of them stop at
// the first non-zero any of them meets, so an array with data costs next to nothing.
bool host_interior_is_zero(const double *v, int m)
{
    if (m < 3) return true;
    const int T = io_threads();
    std::atomic<bool> nonzero{false};
    auto scan = [&](int t) {
        const size_t ncol = (size_t)m - 2;
        const size_t j0 = 1 + ncol * (size_t)t / (size_t)T, j1 = 1 + ncol * (size_t)(t + 1) / (size_t)T;
        for (size_t j = j0; j < j1 && !nonzero.load(std::memory_order_relaxed); j++) {
            const uint64_t *c = reinterpret_cast<const uint64_t *>(v + j * (size_t)m);
            uint64_t acc = 0;
            for (int i = 1; i < m - 1; i++) acc |= c[i];
            if (acc) nonzero.store(true, std::memory_order_relaxed);
        }
    };
    run_tasks(T, scan);
    return !nonzero.load();
}

} // namespace mgx
