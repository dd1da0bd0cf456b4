// tmpfs_falloc.cpp -- how fast does fallocate() give one tmpfs file its pages (one call per 256 MB, 1-4 threads on
// disjoint ranges), and how fast do T threads then fill the mapped pages.   g++ -O2 -pthread tmpfs_falloc.cpp
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{ const size_t GB = (size_t)1 << 30, total = (argc > 1 ? atoll(argv[1]) : 3)*GB;
  const char *dir = argc > 2 ? argv[2] : "/dev/shm";
  std::vector<char> src(64 << 20, 'A');
  for (int T : { 1, 2, 4 })
    { std::string f = std::string(dir)+"/tf_one";
      int fd = open(f.c_str(),O_RDWR|O_CREAT|O_TRUNC,0644);
      if (ftruncate(fd,total)) return 1;
      double t0 = now();
      std::vector<std::thread> th;
      const size_t per = total/T, chunk = (size_t)256 << 20;
      for (int t = 0; t < T; t++) th.emplace_back([&,t] { for (size_t o = 0; o < per; o += chunk) if (fallocate(fd,0,t*per+o,std::min(chunk,per-o))) { perror("fallocate"); exit(1); } });
      for (auto &x : th) x.join();
      double dt = now()-t0;
      printf("fallocate, %d thread(s): %6.2f GB/s\n",T,total/dt/1e9); fflush(stdout);
      char *a = (char *)mmap(nullptr,total,PROT_READ|PROT_WRITE,MAP_SHARED,fd,0);
      for (int W : { 1, 4, 16 })
        { t0 = now();
          std::vector<std::thread> tw;
          const size_t pw = total/W;
          for (int t = 0; t < W; t++) tw.emplace_back([&,t] { for (size_t o = 0; o < pw; o += src.size()) memcpy(a+t*pw+o,src.data(),std::min(src.size(),pw-o)); });
          for (auto &x : tw) x.join();
          printf("   fill mapped pages, %2d thread(s): %6.2f GB/s\n",W,total/(now()-t0)/1e9); fflush(stdout);
        }
      munmap(a,total); close(fd); unlink(f.c_str());
    }
  return 0;
}
