// thread_pool.h -- the host threads of the command line (`-T`, ClassPro.c:530,574-578: the reference spawns T
// workers over contiguous read ranges; here the T threads serve every data-parallel host loop of the pipeline:
// input indexing, staging copies into pinned memory, record formatting).  parallel_for blocks the caller,
// which takes part in the work, so nested use from several pipeline stages cannot dead-lock.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <deque>

class ThreadPool
  { struct Job
      { std::function<void(int64_t)> fn;
        int64_t n = 0;
        std::atomic<int64_t> next{0}, done{0};
      };
    std::vector<std::thread> workers;
    std::deque<std::shared_ptr<Job>> jobs;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    bool quit = false;

    static void run(Job &j, std::condition_variable &cvd, std::mutex &m)
    { for (;;)
        { const int64_t i = j.next.fetch_add(1);
          if (i >= j.n) break;
          j.fn(i);
          if (j.done.fetch_add(1)+1 == j.n)
            { std::lock_guard<std::mutex> lk(m); cvd.notify_all(); }
        }
    }
    void loop()
    { for (;;)
        { std::shared_ptr<Job> j;
          { std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk,[&] { return quit || !jobs.empty(); });
            if (jobs.empty()) return;
            j = jobs.front();
            if (j->next.load() >= j->n) { jobs.pop_front(); continue; }
          }
          run(*j,cv_done,mu);
        }
    }
  public:
    explicit ThreadPool(int nthreads)
    { for (int t = 1; t < nthreads; t++) workers.emplace_back([this] { loop(); }); }
    ~ThreadPool()
    { { std::lock_guard<std::mutex> lk(mu); quit = true; }
      cv.notify_all();
      for (auto &w : workers) w.join();
    }
    int size() const { return (int)workers.size()+1; }

    // fn(i) for i in [0,n), spread over the pool; returns when all are done
    void parallel_for(int64_t n, const std::function<void(int64_t)> &fn)
    { if (n <= 0) return;
      if (n == 1 || workers.empty()) { for (int64_t i = 0; i < n; i++) fn(i); return; }
      auto j = std::make_shared<Job>();
      j->fn = fn; j->n = n;
      { std::lock_guard<std::mutex> lk(mu); jobs.push_back(j); }
      cv.notify_all();
      run(*j,cv_done,mu);
      std::unique_lock<std::mutex> lk(mu);
      cv_done.wait(lk,[&] { return j->done.load() >= j->n; });
      for (auto it = jobs.begin(); it != jobs.end(); ++it)
        if (*it == j) { jobs.erase(it); break; }
    }
  };
