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
