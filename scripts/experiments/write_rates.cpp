// What the page cache takes: ways of getting 8 GB from a host buffer into files under a directory (default /dev/shm).
//   g++ -O2 -pthread scripts/experiments/write_rates.cpp -o /tmp/write_rates && /tmp/write_rates [dir] [GB]
// Behind store.cpp's choice of drain (profiles/r03_write_rates.txt).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void wr(int fd, const char* p, size_t n, size_t at) {
    while (n) { ssize_t w = pwrite(fd, p, n, at); if (w <= 0) { perror("pwrite"); exit(1); } p += w; n -= w; at += w; }
}
int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm";
    const size_t total = (size_t)(atof(argc > 2 ? argv[2] : "8") * 1e9) & ~(size_t)((64 << 20) - 1);
    const size_t PIECE = 64 << 20;
    char* src = (char*)malloc(PIECE);
    memset(src, 'A', PIECE);
    auto path = [&](int k) { return dir + "/bsk_wr_probe." + std::to_string(k); };
    auto report = [&](const char* what, double s) { printf("%-72s %6.2f s  %6.2f GB/s\n", what, s, total / s / 1e9); fflush(stdout); };
    {   // (a) one thread, one file
        int fd = open(path(0).c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        for (size_t at = 0; at < total; at += PIECE) wr(fd, src, PIECE, at);
        report("a  1 thread, pwrite of 64 MiB pieces, one file", now() - t0);
        close(fd); unlink(path(0).c_str());
    }
    for (int K : {2, 4, 8, 16}) {  // (b) K threads, disjoint ranges of ONE file
        int fd = open(path(0).c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        std::vector<std::thread> th;
        for (int k = 0; k < K; ++k) th.emplace_back([&, k] { for (size_t i = k; i * PIECE < total; i += K) wr(fd, src, PIECE, i * PIECE); });
        for (auto& t : th) t.join();
        char b[96]; snprintf(b, sizeof b, "b  %2d threads, pwrite, disjoint pieces of ONE file", K);
        report(b, now() - t0);
        close(fd); unlink(path(0).c_str());
    }
    for (int K : {2, 4, 8, 16, 32}) {  // (c) K threads, a file each
        double t0 = now();
        std::vector<std::thread> th;
        for (int k = 0; k < K; ++k) th.emplace_back([&, k] {
            int fd = open(path(k).c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
            size_t at = 0;
            for (size_t i = k; i * PIECE < total; i += K) { wr(fd, src, PIECE, at); at += PIECE; }
            close(fd);
        });
        for (auto& t : th) t.join();
        char b[96]; snprintf(b, sizeof b, "c  %2d threads, pwrite, a file each", K);
        report(b, now() - t0);
        for (int k = 0; k < K; ++k) unlink(path(k).c_str());
    }
    {   // (d) fallocate first (timed apart), then one thread pwrite
        int fd = open(path(0).c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        int rc = posix_fallocate(fd, 0, total);
        double t1 = now();
        for (size_t at = 0; at < total; at += PIECE) wr(fd, src, PIECE, at);
        double t2 = now();
        char b[96]; snprintf(b, sizeof b, "d  posix_fallocate (rc %d) alone", rc);
        report(b, t1 - t0);
        report("d  1 thread pwrite into the preallocated file", t2 - t1);
        close(fd); unlink(path(0).c_str());
    }
    for (int K : {1, 4, 16}) {  // (e) ftruncate + mmap + K threads memcpy
        int fd = open(path(0).c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        if (ftruncate(fd, total)) perror("ftruncate");
        char* m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) { perror("mmap"); return 1; }
        std::vector<std::thread> th;
        for (int k = 0; k < K; ++k) th.emplace_back([&, k] { for (size_t i = k; i * PIECE < total; i += K) memcpy(m + i * PIECE, src, PIECE); });
        for (auto& t : th) t.join();
        munmap(m, total);
        char b[96]; snprintf(b, sizeof b, "e  %2d threads, memcpy into a shared mapping of one file", K);
        report(b, now() - t0);
        close(fd); unlink(path(0).c_str());
    }
    return 0;
}
