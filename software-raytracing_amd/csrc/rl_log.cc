// Logging for the MI355X raylib: LOG-style printf queue drained by a background
// thread, as the reference's core/logger.cc does (queue + 100 ms drain), minus the
// MSVC CRT.  RAYLIB_QUIET=1 silences it.  The state is heap-allocated and never
// destroyed, and the drain thread is detached, so a process that exits without
// Raylib_Terminate (the C# GUI does) shuts down cleanly.
#include "rl_host.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <thread>

namespace rl {
namespace {
struct LogState {
	std::mutex mu;
	std::condition_variable cv;
	std::deque<std::string> queue;
	bool running = false, stop = false, busy = false;
};
LogState& St() { static LogState* s = new LogState; return *s; }
bool Quiet() { static int q = -1; if (q < 0) { const char* e = getenv("RAYLIB_QUIET"); q = (e && e[0] == '1') ? 1 : 0; } return q == 1; }

void Drain()
{
	LogState& S = St();
	std::unique_lock<std::mutex> lk(S.mu);
	for (;;) {
		S.cv.wait_for(lk, std::chrono::milliseconds(100), [&] { return S.stop || !S.queue.empty(); });
		S.busy = true;
		while (!S.queue.empty()) {
			std::string s = std::move(S.queue.front()); S.queue.pop_front();
			lk.unlock(); fputs(s.c_str(), stdout); fputc('\n', stdout); lk.lock();
		}
		fflush(stdout);
		S.busy = false;
		S.cv.notify_all();
		if (S.stop) break;
	}
	S.running = false;
	S.cv.notify_all();
}
} // namespace

void Log(const char* fmt, ...)
{
	if (Quiet()) return;
	char buf[1024];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
	LogState& S = St();
	std::lock_guard<std::mutex> lk(S.mu);
	if (!S.running) { fputs(buf, stdout); fputc('\n', stdout); fflush(stdout); return; }
	S.queue.emplace_back(buf);
	S.cv.notify_all();
}

void LogStart()
{
	LogState& S = St();
	std::lock_guard<std::mutex> lk(S.mu);
	if (S.running) return;
	S.stop = false; S.running = true;
	std::thread(Drain).detach();
}

void LogFlush()
{
	LogState& S = St();
	std::unique_lock<std::mutex> lk(S.mu);
	if (!S.running) { fflush(stdout); return; }
	S.cv.notify_all();
	S.cv.wait_for(lk, std::chrono::seconds(5), [&] { return S.queue.empty() && !S.busy; });
}

void LogStop()
{
	LogState& S = St();
	std::unique_lock<std::mutex> lk(S.mu);
	if (!S.running) return;
	S.stop = true;
	S.cv.notify_all();
	S.cv.wait_for(lk, std::chrono::seconds(5), [&] { return !S.running; });
}

} // namespace rl
