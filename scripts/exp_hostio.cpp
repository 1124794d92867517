// How fast can compressed trajectory bytes get from the page cache to the GPU?  (round 3, file-backed XTC path)
//   a) pread into one pinned buffer on T threads, then one DMA   (what the evaluator's staging does)
//   b) mmap the file, hipHostRegister the mapping, DMA straight from the page cache (no CPU copy) - if the driver allows it
// build: hipcc -O2 -o build/exp_hostio scripts/exp_hostio.cpp -lpthread ; run: build/exp_hostio [MB]
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t MB = argc > 1 ? (size_t)atoi(argv[1]) : 512;
    const size_t bytes = MB << 20;
    const char* path = "/tmp/exp_hostio.bin";
    {   // a file in the page cache
        std::vector<char> buf(1 << 20);
        for (size_t i = 0; i < buf.size(); ++i) buf[i] = (char)(i * 131);
        FILE* f = fopen(path, "wb");
        for (size_t m = 0; m < MB; ++m) fwrite(buf.data(), 1, buf.size(), f);
        fclose(f);
    }
    int fd = open(path, O_RDONLY);
    void* pin = nullptr;
    void* dev = nullptr;
    if (hipHostMalloc(&pin, bytes, hipHostMallocDefault) != hipSuccess || hipMalloc(&dev, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipStream_t s;
    hipStreamCreate(&s);
    const size_t piece = 512 << 10;          // one compressed c2 frame
    for (int T : {1, 4, 8, 16, 32, 64}) {
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            std::atomic<size_t> next{0};
            const double t0 = now();
            auto work = [&]() {
                for (;;) {
                    const size_t o = next.fetch_add(piece);
                    if (o >= bytes) break;
                    size_t n = std::min(piece, bytes - o), done = 0;
                    while (done < n) { ssize_t r = pread(fd, (char*)pin + o + done, n - done, (off_t)(o + done)); if (r <= 0) break; done += (size_t)r; }
                }
            };
            std::vector<std::thread> pool;
            for (int t = 1; t < T; ++t) pool.emplace_back(work);
            work();
            for (auto& t : pool) t.join();
            best = std::min(best, now() - t0);
        }
        printf("pread into pinned memory, %2d threads: %6.1f ms for %zu MB = %5.1f GB/s\n", T, best * 1e3, MB, bytes / best / 1e9);
    }
    {
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            const double t0 = now();
            hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s);
            hipStreamSynchronize(s);
            best = std::min(best, now() - t0);
        }
        printf("DMA pinned -> device: %6.1f ms = %5.1f GB/s\n", best * 1e3, bytes / best / 1e9);
    }
    for (int shared = 0; shared < 2; ++shared) {
        void* map = mmap(nullptr, bytes, PROT_READ, (shared ? MAP_SHARED : MAP_PRIVATE) | MAP_POPULATE, fd, 0);
        if (map == MAP_FAILED) { printf("mmap failed\n"); continue; }
        for (unsigned flags : {(unsigned)hipHostRegisterDefault, (unsigned)hipHostRegisterReadOnly}) {
            const double t0 = now();
            hipError_t e = hipHostRegister(map, bytes, flags);
            const double t1 = now();
            printf("hipHostRegister(mmap %s, flags %u): %s in %.1f ms\n", shared ? "MAP_SHARED" : "MAP_PRIVATE", flags, hipGetErrorString(e), (t1 - t0) * 1e3);
            if (e != hipSuccess) { (void)hipGetLastError(); continue; }
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                const double a = now();
                hipError_t c = hipMemcpyAsync(dev, map, bytes, hipMemcpyHostToDevice, s);
                hipStreamSynchronize(s);
                best = std::min(best, now() - a);
                if (c != hipSuccess) printf("  copy: %s\n", hipGetErrorString(c));
            }
            printf("  DMA from the registered mapping: %6.1f ms = %5.1f GB/s\n", best * 1e3, bytes / best / 1e9);
            std::vector<char> chk(4096);
            hipMemcpy(chk.data(), (char*)dev + (bytes - 4096), 4096, hipMemcpyDeviceToHost);
            printf("  content %s\n", memcmp(chk.data(), (char*)map + (bytes - 4096), 4096) == 0 ? "ok" : "MISMATCH");
            hipHostUnregister(map);
        }
        munmap(map, bytes);
    }
    close(fd);
    unlink(path);
    return 0;
}
