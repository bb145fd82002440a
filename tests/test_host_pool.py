"""The library's host pool (modkit_amd/csrc/mkp_bam.hpp: HostPool): sized from the CPUs the process may use (affinity mask, cgroup quota),
overridable with MKP_POOL_THREADS, and the same bedMethyl plan whatever its size."""
import os
import re
import subprocess
import sys

import pytest

import modkit_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pool_threads(env=None, taskset=None):
    code = "import sys; sys.path.insert(0, %r); import modkit_amd; print(modkit_amd.lib().mkp_host_threads())" % ROOT
    cmd = [sys.executable, "-c", code]
    if taskset:
        cmd = ["taskset", "-c", taskset] + cmd
    return int(subprocess.check_output(cmd, env=dict(os.environ, **(env or {}))).decode().split()[-1])


def usable_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def test_pool_follows_the_usable_cpus():
    n = pool_threads()
    assert n == min(64, usable_cpus()) >= 1
    assert modkit_amd.lib().mkp_host_threads() == n


def test_pool_override_and_affinity():
    assert pool_threads({"MKP_POOL_THREADS": "3"}) == 3
    assert pool_threads({"MKP_POOL_THREADS": "0"}) == 1          # clamped
    if len(os.sched_getaffinity(0)) >= 2 and subprocess.call(["which", "taskset"], stdout=subprocess.DEVNULL) == 0:
        first = sorted(os.sched_getaffinity(0))[0]
        assert pool_threads(taskset=str(first)) == 1             # one CPU in the mask: one thread


@pytest.mark.parametrize("threads", ["1", "2", "5"])
def test_plan_and_pack_do_not_depend_on_the_pool_size(tmp_path, threads):
    """--plan-only prints per shard the packer's digest of everything it would hand to the device: the same for any pool size (work is
    split by the pool, foreground jobs ahead of the prefetch; the result may not move)."""
    from test_host_ingest import gen
    bam, _ = gen(tmp_path, "pp", [("c0", 300_000), ("c1", 200_000)], 9000, ["--mean-len", "3000"])
    cli = os.path.join(os.path.dirname(modkit_amd.LIB_PATH), "mkpileup")
    modkit_amd.build()
    outs = []
    for t in ("", threads):
        out = str(tmp_path / ("plan%s.tsv" % t))
        env = dict(os.environ)
        if t:
            env.update(MKP_POOL_THREADS=t, MKP_PACK_PIECES=t)
        p = subprocess.run([cli, "pileup", bam, out, "--plan-only", "--shard-bp", "100000", "--stats"], capture_output=True, text=True, env=env)
        assert p.returncode == 0, p.stderr
        assert re.search(r"shards=\d+", p.stderr)
        outs.append(open(out).read())
    assert outs[0] == outs[1] and outs[0].count("\n") >= 5


POOL_SRC = r"""
#include <atomic>
#include <chrono>
#include <cstdio>
#include <future>
#include <mutex>
#include <set>
#include <thread>
#include "mkp_bam.hpp"
using namespace mkp;
static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL line %d: %s\n", __LINE__, #c); fails++; } } while (0)
int main() {
  HostPool& P = HostPool::get();
  CHECK(P.size() == 4);   // MKP_POOL_THREADS=4 from the test
  // every index exactly once, from several submitting threads at a time
  { std::vector<std::atomic<int>> hit(10000); for (auto& h : hit) h = 0;
    auto job = [&](size_t lo) { P.parallel(2500, [&](size_t i) { hit[lo + i]++; }); };
    std::thread a(job, 0), b(job, 2500), c(job, 5000); job(7500); a.join(); b.join(); c.join();
    for (auto& h : hit) CHECK(h == 1); }
  // an exception in a task reaches the caller after the job is retired; the pool keeps working
  { bool threw = false; try { P.parallel(64, [&](size_t i) { if (i == 13) throw Error(MKP_E_IO, "boom"); }); } catch (const Error& e) { threw = e.status == MKP_E_IO; } CHECK(threw);
    std::atomic<int> n{0}; P.parallel(100, [&](size_t) { n++; }); CHECK(n == 100); }
  // foreground before background: with a long background job queued, a foreground job submitted later still finishes first
  { std::atomic<bool> bg_done{false}; std::atomic<int> bg_ran{0};
    auto bg = std::async(std::launch::async, [&]() { HostPool::background() = true;
        P.parallel(400, [&](size_t) { std::this_thread::sleep_for(std::chrono::milliseconds(1)); bg_ran++; }); bg_done = true; });
    while (bg_ran < 8) std::this_thread::yield();            // the background job has the workers
    std::atomic<int> fg{0}; std::mutex mu; std::set<std::thread::id> who;
    P.parallel(40, [&](size_t) { { std::lock_guard<std::mutex> g(mu); who.insert(std::this_thread::get_id()); } std::this_thread::sleep_for(std::chrono::milliseconds(1)); fg++; });
    CHECK(fg == 40); CHECK(!bg_done);                          // ~130 ms of background work on the pool is not over ...
    CHECK(who.size() >= 2);                                    // ... yet workers took foreground tasks (queued behind it, only the caller would have)
    bg.get(); CHECK(bg_ran == 400); }
  // inflate windows are recycled: a released large buffer comes back for a request of a similar size, and trim gives it up
  { ByteBuf a; a.alloc((size_t)70 << 20); uint8_t* p = a.data(); a[0] = 1; a.release();
    ByteBuf b; b.alloc((size_t)66 << 20); CHECK(b.data() == p); b.release();
    ByteBuf c; c.alloc((size_t)200 << 20); CHECK(c.data() != p);   // nothing parked fits: the parked one is returned to the system
    c.release(); ByteBuf::trim_spares(); ByteBuf d; d.alloc((size_t)200 << 20); d[0] = 2; }
  printf(fails ? "FAILED %d\n" : "ok\n", fails);
  return fails ? 1 : 0;
}
"""


def test_pool_priority_exceptions_and_buffer_recycling(tmp_path):
    src = tmp_path / "pool.cpp"
    src.write_text(POOL_SRC)
    exe = tmp_path / "pool"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "modkit_amd", "csrc"), "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src), "-lz", "-pthread"])
    p = subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, MKP_POOL_THREADS="4"))
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stdout + p.stderr
