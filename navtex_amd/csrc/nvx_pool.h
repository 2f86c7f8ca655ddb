// nvx_pool.h -- the persistent host workers of one handle (internal).
// The character layers of a collected launch (nvx_sitor.c, one per chain: receiver/nav_b_sm.C) run side by side on up
// to 16 threads.  Creating and joining those threads per collect cost ~0.1 ms per launch at 50 launches a second; the
// pool starts them once, on first use, and parks them on a condition variable between collects.
#ifndef NVX_POOL_H
#define NVX_POOL_H

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

struct HostPool {
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable go, idle;
    const std::function<void(int)> *job = nullptr;
    int n_job = 0, next = 0, pending = 0;
    unsigned long long generation = 0;
    bool quit = false;

    // fn(t) for t = 0 .. n-1, each exactly once, on the workers and the calling thread; returns when all have finished.
    // One run at a time (the handle is locked by its caller).
    void run(int n, const std::function<void(int)> &fn)
    {
        if (n <= 1) { if (n == 1) fn(0); return; }
        while ((int)workers.size() < n - 1) workers.emplace_back([this] { loop(); });
        {
            std::lock_guard<std::mutex> lk(mu);
            job = &fn; n_job = n; next = 0; pending = n; generation++;
        }
        go.notify_all();
        take();                                            // the caller works too
        std::unique_lock<std::mutex> lk(mu);
        idle.wait(lk, [&] { return pending == 0; });
        job = nullptr;
    }

    ~HostPool()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        go.notify_all();
        for (auto &t : workers) t.join();
    }

private:
    void take()
    {
        for (;;) {
            int t;
            const std::function<void(int)> *f;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!job || next >= n_job) return;
                t = next++; f = job;
            }
            (*f)(t);
            bool last;
            { std::lock_guard<std::mutex> lk(mu); last = --pending == 0; }
            if (last) idle.notify_all();
        }
    }
    void loop()
    {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                go.wait(lk, [&] { return quit || generation != seen; });
                if (quit) return;
                seen = generation;
            }
            take();
        }
    }
};

#endif
