// file_write.cpp -- how fast can demultiplexed output be appended on this box's file system?  (tools/grun.sh ... './fw DIR')
//   a) pwrite, one thread per file, F files                  (buffered writes to one file serialise on its inode lock)
//   b) pwrite, T threads on disjoint ranges of ONE file
//   c) ftruncate + mmap(MAP_SHARED) + memcpy, T threads on disjoint ranges of one file, F files in turn
// g++ -O2 -pthread -o fw tools/ubench/file_write.cpp
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
	const std::string dir = argc > 1 ? argv[1] : "/tmp";
	const size_t per_file = (size_t)(argc > 2 ? atol(argv[2]) : 256) << 20;
	const int F = 9;
	std::vector<char> src(per_file);
	for (size_t i = 0; i < per_file; i++) src[i] = (char)('A' + i % 23);
	auto name = [&](const char* tag, int f) { return dir + "/fw_" + tag + std::to_string(f) + ".bin"; };
	{   // a
		double t0 = now();
		std::vector<std::thread> th;
		for (int f = 0; f < F; f++) th.emplace_back([&, f] {
			int fd = open(name("a", f).c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
			for (size_t o = 0; o < per_file; o += 1 << 20) if (pwrite(fd, src.data() + o, 1 << 20, (off_t)o) < 0) perror("pwrite");
			close(fd);
		});
		for (auto& t : th) t.join();
		double dt = now() - t0;
		printf("a) pwrite, %d files x 1 thread each: %.2f GB/s\n", F, F * per_file / dt / 1e9);
	}
	for (int T : { 1, 4, 16 }) {   // b
		double t0 = now();
		int fd = open(name("b", T).c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
		std::vector<std::thread> th;
		const size_t part = per_file / T;
		for (int t = 0; t < T; t++) th.emplace_back([&, t] {
			for (size_t o = t * part; o < (t + 1) * part; o += 1 << 20) if (pwrite(fd, src.data() + o, 1 << 20, (off_t)o) < 0) perror("pwrite");
		});
		for (auto& t : th) t.join();
		close(fd);
		printf("b) pwrite, one file, %d threads: %.2f GB/s\n", T, per_file / (now() - t0) / 1e9);
	}
	for (int T : { 1, 8, 16, 32 }) {   // c
		double t0 = now();
		for (int f = 0; f < F; f++) {
			int fd = open(name("c", f).c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
			if (ftruncate(fd, (off_t)per_file) != 0) perror("ftruncate");
			char* m = (char*)mmap(nullptr, per_file, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
			if (m == MAP_FAILED) { perror("mmap"); return 1; }
			std::vector<std::thread> th;
			const size_t part = per_file / T;
			for (int t = 0; t < T; t++) th.emplace_back([&, t] { memcpy(m + t * part, src.data() + t * part, part); });
			for (auto& t : th) t.join();
			munmap(m, per_file);
			close(fd);
		}
		printf("c) mmap + memcpy, %d files in turn, %d threads: %.2f GB/s\n", F, T, F * per_file / (now() - t0) / 1e9);
	}
	for (int f = 0; f < 40; f++) for (const char* tag : { "a", "b", "c" }) unlink(name(tag, f).c_str());
	return 0;
}
