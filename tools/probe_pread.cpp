// Development probe (any Linux box): how fast do N threads get a just-written file's bytes into an anonymous buffer -
// pread (first and second pass over the file) against a copy off a shared read-only mapping (first and second pass)?
// g++ -O2 -pthread tools/probe_pread.cpp -o /tmp/probe_pread && /tmp/probe_pread /dev/shm/x.bin 4096 32
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
template <class F> static double par(int threads, size_t pieces, F f) {
    std::atomic<size_t> next(0);
    const double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back([&] { for (size_t i; (i = next++) < pieces;) f(i); });
    for (auto& x : th) x.join();
    return now() - t0;
}
int main(int argc, char** argv) {
    const char* path = argv[1];
    const size_t mb = argc > 2 ? atol(argv[2]) : 4096;
    const int threads = argc > 3 ? atoi(argv[3]) : 32;
    const size_t piece = (argc > 4 ? atol(argv[4]) : 1024) << 10, bytes = mb << 20, pieces = bytes / piece;
    char* buf = (char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    memset(buf, 1, bytes);
    for (int round = 0; round < 2; ++round) {
        const char* how = round == 0 ? "pread" : "mmap copy";
        unlink(path);
        int fd = open(path, O_CREAT | O_RDWR, 0600);
        double t = par(threads, pieces, [&](size_t i) { if (pwrite(fd, buf + i * piece, piece, i * piece) != (ssize_t)piece) abort(); });
        printf("%-9s: written %zu MB with %d threads in %.3f s (%.1f GB/s)\n", how, mb, threads, t, bytes / t / 1e9);
        close(fd);
        fd = open(path, O_RDONLY);
        const char* map = round ? (const char*)mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0) : nullptr;
        for (int pass = 0; pass < 3; ++pass) {
            t = par(threads, pieces, [&](size_t i) {
                if (round == 0) { size_t o = 0; while (o < piece) { ssize_t g = pread(fd, buf + i * piece + o, piece - o, i * piece + o); if (g <= 0) abort(); o += g; } }
                else memcpy(buf + i * piece, map + i * piece, piece);
            });
            printf("%-9s: pass %d over the file: %.3f s (%.1f GB/s)\n", how, pass, t, bytes / t / 1e9);
        }
        if (map) munmap((void*)map, bytes);
        close(fd);
    }
    unlink(path);
    return 0;
}
