// tmpfs_write.cpp -- what the host can do when T threads write one large output (the .class file): memcpy into
// anonymous memory, into a fresh shared mapping of ONE tmpfs file, pwrite into ONE tmpfs file, pwrite into T
// files, and into a mapping whose pages already exist.   g++ -O2 -pthread tmpfs_write.cpp -o tmpfs_write
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
  for (int T : { 1, 4, 8, 16 })
    { const size_t per = total/T;
      auto run = [&](const char *what, auto fn)
        { double t0 = now();
          std::vector<std::thread> th;
          for (int t = 0; t < T; t++) th.emplace_back([&,t] { fn(t); });
          for (auto &x : th) x.join();
          double dt = now()-t0;
          printf("T=%2d %-34s %6.2f GB/s\n",T,what,total/dt/1e9); fflush(stdout);
        };
      { char *a = (char *)mmap(nullptr,total,PROT_READ|PROT_WRITE,MAP_PRIVATE|MAP_ANONYMOUS,-1,0);
        run("anonymous memory (fresh)",[&](int t) { for (size_t o = 0; o < per; o += src.size()) memcpy(a+t*per+o,src.data(),std::min(src.size(),per-o)); });
        run("anonymous memory (touched)",[&](int t) { for (size_t o = 0; o < per; o += src.size()) memcpy(a+t*per+o,src.data(),std::min(src.size(),per-o)); });
        munmap(a,total);
      }
      std::string f = std::string(dir)+"/tw_one";
      { int fd = open(f.c_str(),O_RDWR|O_CREAT|O_TRUNC,0644);
        if (ftruncate(fd,total)) return 1;
        char *a = (char *)mmap(nullptr,total,PROT_READ|PROT_WRITE,MAP_SHARED,fd,0);
        run("one file, shared mapping (fresh)",[&](int t) { for (size_t o = 0; o < per; o += src.size()) memcpy(a+t*per+o,src.data(),std::min(src.size(),per-o)); });
        run("one file, shared mapping (exists)",[&](int t) { for (size_t o = 0; o < per; o += src.size()) memcpy(a+t*per+o,src.data(),std::min(src.size(),per-o)); });
        munmap(a,total); close(fd); unlink(f.c_str());
      }
      { int fd = open(f.c_str(),O_RDWR|O_CREAT|O_TRUNC,0644);
        run("one file, pwrite 4 MB (fresh)",[&](int t) { for (size_t o = 0; o < per; o += 4 << 20) if (pwrite(fd,src.data(),std::min((size_t)4 << 20,per-o),t*per+o) < 0) exit(1); });
        run("one file, pwrite 4 MB (exists)",[&](int t) { for (size_t o = 0; o < per; o += 4 << 20) if (pwrite(fd,src.data(),std::min((size_t)4 << 20,per-o),t*per+o) < 0) exit(1); });
        close(fd); unlink(f.c_str());
      }
      { std::vector<int> fds;
        for (int t = 0; t < T; t++) fds.push_back(open((f+std::to_string(t)).c_str(),O_RDWR|O_CREAT|O_TRUNC,0644));
        run("T files, pwrite 4 MB (fresh)",[&](int t) { for (size_t o = 0; o < per; o += 4 << 20) if (pwrite(fds[t],src.data(),std::min((size_t)4 << 20,per-o),o) < 0) exit(1); });
        for (int t = 0; t < T; t++) { close(fds[t]); unlink((f+std::to_string(t)).c_str()); }
      }
    }
  return 0;
}
