// host_pool.hpp -- a small persistent pool of host threads for the set-up phases (edge sort/gather, Hsc pattern).
// Spawning std::threads per parallel loop costs about a millisecond per loop inside a process that has the HIP
// runtime mapped (16 clones with large address spaces); the set-up path runs ten such loops.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace cubahip
{

class HostPool
{
public:
	static HostPool& instance()
	{
		static HostPool pool;
		return pool;
	}

	int maxThreads() const { return (int)workers_.size() + 1; }

	// fn(t) for t in [0, T): T - 1 pool workers plus the calling thread; returns when all are done.
	template <class F>
	void run(int T, F&& fn)
	{
		T = std::min(T, maxThreads());
		if (T <= 1) { fn(0); return; }
		std::unique_lock<std::mutex> callers(callerMutex_);       // one parallel region at a time
		{
			std::lock_guard<std::mutex> lk(m_);
			job_ = [&fn](int t) { fn(t); };
			want_ = T - 1; pending_ = T - 1; generation_++;
			hint_.store(generation_, std::memory_order_release);
		}
		start_.notify_all();
		fn(T - 1);
		std::unique_lock<std::mutex> lk(m_);
		done_.wait(lk, [&] { return pending_ == 0; });
		job_ = nullptr;
	}

private:
	HostPool()
	{
		const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
		unsigned cap = 32;                               // (the loops are pointer chases: they scale with the number of outstanding misses, not with flops)
		if (const char* e = std::getenv("CUBA_HIP_HOST_THREADS")) cap = (unsigned)std::max(1, std::atoi(e));       // A/B knob
		const int n = (int)std::min(cap, hw) - 1;
		for (int w = 0; w < n; w++) workers_.emplace_back([this, w] { loop(w); });
	}
	~HostPool()
	{
		{
			std::lock_guard<std::mutex> lk(m_);
			stop_ = true;
		}
		start_.notify_all();
		for (auto& t : workers_) t.join();
	}
	void loop(int w)
	{
		unsigned long seen = 0;
		for (;;)
		{
			std::function<void(int)> job;
			// parallel regions come in bursts (initialize() + set_graph run five in a row): a worker that has just finished one spins
			// for a moment before it goes back to sleep, so that the next region of the burst does not pay a futex wake-up per thread
#ifndef CUBA_HIP_POOL_NO_SPIN
			if (seen != 0)
			{
				const auto t0 = std::chrono::steady_clock::now();
				for (int spins = 0; hint_.load(std::memory_order_acquire) == seen; spins++)
				{
					if ((spins & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
#if defined(__x86_64__) || defined(__i386__)
					__builtin_ia32_pause();
#endif
				}
			}
#endif
			{
				std::unique_lock<std::mutex> lk(m_);
				start_.wait(lk, [&] { return stop_ || generation_ != seen; });
				if (stop_) return;
				seen = generation_;
				if (w >= want_) continue;
				job = job_;
			}
			job(w);
			{
				std::lock_guard<std::mutex> lk(m_);
				if (--pending_ == 0) done_.notify_all();
			}
		}
	}

	std::vector<std::thread> workers_;
	std::mutex m_, callerMutex_;
	std::condition_variable start_, done_;
	std::function<void(int)> job_;
	int want_ = 0, pending_ = 0;
	unsigned long generation_ = 0;
	std::atomic<unsigned long> hint_{ 0 };     // copy of generation_ the spinning workers watch without the mutex
	bool stop_ = false;
};

}  // namespace cubahip
