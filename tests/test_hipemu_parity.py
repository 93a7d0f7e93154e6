"""The kernel SOURCES on the host wavefront emulator (tools/hipemu: fibers per lane, cross-lane operations as rendezvous) against the
oracle — bit for bit, on a machine without a GPU.  Test infrastructure on both sides: the emulator library is built from
maelstrom_amd/csrc by tools/hipemu/build_emu.py with the host compiler and loaded through MSIM_LIB in a child process; the product
library (hipcc, gfx950) is not involved and still refuses to run without a device.  One small case per kernel layout the round touched:
the two-clusters-per-wavefront broadcast kernel (constant and random latency), the wide kernel with the nodes' sets in LDS and its
lone-operation path, eight clusters per wavefront for two txn-list-append nodes, the Datomic-style one (one cluster per wavefront), for txn-rw-register, echo / unique-ids, g-set / the counters the broadcast programs and kafka (one cluster per wavefront; its committed-offset lookup), the list-append check's workgroup-per-history kernel, the kafka checker's device pass."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tools", "hipemu", "_build", "libmaelsim_emu.so")

CASES = [
    "duo25", "duo25exp",
    "{'workload':'g-set','node_count':40,'rate':40,'time_limit':6,'latency':50,'latency_dist':'exponential','p_loss':0.05,'n':2}",
    "{'workload':'g-set','node_count':40,'rate':40,'time_limit':11,'latency':50,'latency_dist':'exponential','n':2,'flags':0x4000}",   # whole ticks: the union merge; sets in LDS
    "{'workload':'g-set','node_count':45,'rate':50,'time_limit':12,'latency':3000,'latency_dist':'exponential','n':1}",             # ticks overlap: a slot is flushed for another tick
    "{'workload':'pn-counter','node_count':40,'rate':50,'time_limit':12,'latency':50,'p_loss':0.2,'n':1}",
    "{'workload':'pn-counter','node_count':50,'rate':100,'time_limit':11,'latency':200,'latency_dist':'exponential','n':1}",   # quiet windows: a time jump past the whole per-millisecond table
    "{'workload':'g-set','node_count':100,'rate':100,'time_limit':11,'latency':100,'latency_dist':'exponential','n':1}",       # quiet windows at cfg3's shape
    "{'workload':'broadcast','node_count':36,'rate':20,'time_limit':3,'latency':10,'topology':'tree3','n':1}",
    "{'workload':'txn-list-append','bin':'multi-key-txn','node_count':5,'rate':60,'time_limit':5,'latency':5,'nemesis':['partition'],'nemesis_interval':2,'n':9}",
    "{'workload':'txn-list-append','node_count':5,'rate':60,'time_limit':5,'latency':5,'nemesis':['partition'],'nemesis_interval':2,'n':9}",
    "{'workload':'txn-list-append','bin':'datomic','node_count':5,'rate':60,'time_limit':5,'latency':5,'nemesis':['partition'],'nemesis_interval':2,'n':3,'journal_capacity':100000}",   # (journal on: one cluster per wavefront)
    "{'workload':'txn-list-append','bin':'datomic','node_count':3,'rate':150,'time_limit':6,'latency':0,'key_count':16,'max_writes_per_key':2,'n':2}",   # ~600 keys: splits at every level, chains; one cluster per wavefront
    "{'workload':'txn-list-append','bin':'datomic','node_count':3,'rate':150,'time_limit':6,'latency':0,'key_count':16,'max_writes_per_key':2,'n':3,'flags':0x400}",   # the same, eight per wavefront (dt8.hip)
    "{'workload':'txn-list-append','bin':'datomic','node_count':5,'rate':60,'time_limit':5,'latency':5,'nemesis':['partition'],'nemesis_interval':2,'n':11,'flags':0x400}",
    "{'workload':'txn-list-append','bin':'datomic','node_count':6,'rate':120,'time_limit':4,'latency':10,'latency_dist':'uniform','p_loss':0.02,'n':9,'flags':0x400}",
    "{'workload':'txn-list-append','bin':'datomic','node_count':2,'rate':15,'time_limit':60,'latency':1300,'latency_dist':'exponential','seed':91,'n':8,'flags':0x400}",   # instances 2 and 4: a cas served after its sender's await gave up (the request carries its own from / transaction)
    "{'workload':'txn-list-append','bin':'datomic','node_count':2,'rate':15,'time_limit':60,'latency':1300,'latency_dist':'exponential','seed':91,'n':5}",
    "{'workload':'txn-list-append','bin':'datomic','node_count':1,'concurrency':10,'rate':100,'time_limit':6,'latency':2,'n':2}",   # several workers per node: dtg_kernel<> (a lane per endpoint); the lock's waiting queue under load
    "{'workload':'txn-list-append','bin':'datomic','node_count':5,'concurrency':10,'rate':100,'time_limit':6,'latency':5,'nemesis':['partition'],'nemesis_interval':2,'n':2}",
    "{'workload':'txn-list-append','node_count':2,'concurrency':20,'rate':300,'time_limit':5,'latency':3,'latency_dist':'uniform','n':2}",   # several workers per node, single-root node: txng_kernel<>
    "{'workload':'txn-list-append','bin':'multi-key-txn','node_count':5,'concurrency':10,'rate':100,'time_limit':6,'latency':5,'nemesis':['partition'],'nemesis_interval':2,'n':2}",   # ... multi-key node: mkg_kernel<>
    "{'workload':'txn-rw-register','node_count':2,'rate':100,'time_limit':8,'nemesis':['partition'],'nemesis_interval':2,'flags':0x400,'n':11}",
    "{'workload':'txn-rw-register','node_count':5,'concurrency':10,'rate':100,'time_limit':6,'latency':5,'nemesis':['partition'],'nemesis_interval':2,'n':2}",   # several workers per node: hatg_kernel<>
    "{'workload':'txn-rw-register','node_count':4,'rate':200,'time_limit':6,'latency':20,'latency_dist':'exponential','p_loss':0.05,'flags':0x8400,'n':19}",
    "{'workload':'txn-rw-register','node_count':5,'rate':200,'time_limit':8,'latency':5,'nemesis':['partition'],'nemesis_interval':3,'flags':0x400,'n':9}",
    "{'workload':'kafka','node_count':5,'rate':80,'time_limit':8,'latency':5,'nemesis':['partition'],'nemesis_interval':3,'flags':0x400,'n':9}",
    "{'workload':'kafka','node_count':2,'rate':200,'time_limit':4,'latency':2,'key_count':2,'max_writes_per_key':40,'flags':0x400,'n':9}",
    "{'workload':'kafka','node_count':5,'concurrency':10,'rate':100,'time_limit':8,'latency':5,'nemesis':['partition'],'nemesis_interval':3,'n':2}",   # several workers per node: kafkag_kernel<>
    "{'workload':'unique-ids','node_count':3,'rate':500,'time_limit':4,'latency':5,'nemesis':['partition'],'nemesis_interval':2,'flags':0x400,'n':11}",
    "{'workload':'echo','node_count':5,'rate':300,'time_limit':3,'latency':2,'p_loss':0.1,'flags':0x400,'n':9}",
    "{'workload':'g-set','node_count':5,'rate':100,'time_limit':8,'latency':10,'nemesis':['partition'],'nemesis_interval':3,'flags':0x400,'n':11}",
    "{'workload':'pn-counter','node_count':5,'rate':100,'time_limit':12,'latency':50,'latency_dist':'exponential','p_loss':0.1,'flags':0x400,'n':9}",
    "{'workload':'broadcast','bin':'broadcast-ack-retry','node_count':5,'rate':60,'time_limit':8,'latency':10,'nemesis':['partition'],'nemesis_interval':2,'p_loss':0.1,'flags':0x400,'n':9}",
    "{'workload':'broadcast','node_count':5,'rate':100,'time_limit':5,'latency':5,'topology':'line','flags':0x8400,'n':9}",
    "{'workload':'kafka','node_count':5,'rate':150,'time_limit':6,'latency':20,'latency_dist':'exponential','nemesis':['partition'],'nemesis_interval':2,'n':4}",
    "{'workload':'lin-kv','bin':'lin-kv-proxy','proxy_service':'lin-kv','node_count':5,'concurrency':10,'rate':60,'time_limit':6,'latency':5,'nemesis':['partition'],'nemesis_interval':2,'flags':0x400,'n':6}",   # the lin-kv proxy, four clusters per wavefront (svc4.hip): a full 16-lane group, a full wavefront and a partial one
    "{'workload':'lin-kv','bin':'lin-kv-proxy','proxy_service':'lww-kv','node_count':3,'concurrency':6,'rate':200,'time_limit':4,'latency':20,'latency_dist':'exponential','p_loss':0.05,'flags':0x400,'n':5}",
    "{'workload':'unique-ids','bin':'tso-ids','node_count':3,'concurrency':6,'rate':300,'time_limit':4,'latency':5,'nemesis':['partition'],'nemesis_interval':1,'flags':0x400,'n':5}",   # unique-ids over lin-tso, four clusters per wavefront (svc4_kernel<.., TSO>)
    "{'workload':'txn-list-append','node_count':1,'concurrency':10,'rate':100,'time_limit':6,'latency':5,'flags':0x400,'n':6}",   # the single-root txn node with several workers per node, four clusters per wavefront (txng4.hip)
    "{'workload':'txn-list-append','node_count':5,'concurrency':10,'rate':200,'time_limit':4,'latency':5,'nemesis':['partition'],'nemesis_interval':1,'p_loss':0.05,'flags':0x400,'n':5}",   # a full 16-lane group; lost replies leave slots taken
    "{'workload':'txn-list-append','bin':'datomic','node_count':1,'concurrency':10,'rate':100,'time_limit':6,'latency':2,'flags':0x400,'n':6}",   # the Datomic-style node with several workers per node, four clusters per wavefront (dtg4.hip): the lock's waiting queue under load
    "{'workload':'txn-list-append','bin':'datomic','node_count':2,'concurrency':12,'rate':200,'time_limit':5,'latency':5,'nemesis':['partition'],'nemesis_interval':2,'p_loss':0.03,'flags':0x400,'n':5}",   # a full 16-lane group
    "{'workload':'lin-kv','bin':'lin-kv-proxy','proxy_service':'seq-kv','node_count':3,'rate':100,'time_limit':4,'latency':10,'n':2}",   # seq-kv: one cluster per wavefront (svc_kernel<>)
]


@pytest.fixture(scope="module")
def emu_lib():
    if shutil.which(os.environ.get("HIPEMU_CXX", "g++")) is None:
        pytest.skip("no host C++ compiler for the emulator build")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hipemu", "build_emu.py")], cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(EMU)
    return EMU


@pytest.mark.timeout(1800)
def test_kernel_sources_on_the_emulator_equal_the_oracle(emu_lib):
    # MSIM_GUARD=3: every device slab between two pattern-filled red zones (csrc/guard.cpp), verified when it is freed and at the end
    env = dict(os.environ, MSIM_LIB=emu_lib, HIPEMU_DIVERGENT="1", MSIM_GUARD="3")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "emu_compare.py")] + CASES, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count(": OK") == len(CASES), r.stdout
    assert "guard: 0 damaged byte(s)" in r.stdout, r.stdout[-2000:]


@pytest.mark.timeout(1800)
def test_list_append_check_kernels_on_the_emulator_equal_the_host_analysis(emu_lib):
    """tests/test_txn_check_gpu.py's hand-made anomalies and corrupted histories through the emulated device pass (both kernels)."""
    for flags in ("0", "0x2000"):
        env = dict(os.environ, MSIM_LIB=emu_lib, HIPEMU_DIVERGENT="1", MSIM_DEV_FLAGS=flags, MSIM_TXN_WG="256")
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_txn_check_gpu.py"), "-k", "hand_made"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.timeout(1800)
def test_kafka_check_kernel_on_the_emulator_equals_the_host_checker(emu_lib):
    """tests/test_kafka_check_gpu.py (clean oracle histories, the anomalies by hand, corrupted histories, the engine's check) through the
    emulated device pass of csrc/kafka_check_dev.hip."""
    env = dict(os.environ, MSIM_LIB=emu_lib, HIPEMU_DIVERGENT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_kafka_check_gpu.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.timeout(1800)
def test_device_checkers_on_the_emulator_equal_the_host_checkers(emu_lib):
    """The unique-ids check (a history's id table in LDS / in HBM workspace, tests/test_checker_gpu.py), the rw-register analysis
    (tests/test_rw_check_gpu.py), the wide keys of the lin-kv search (tests/test_lin_check_gpu.py) and the reference-held set-full / linearizability vectors (tests/test_checker_reference_vectors.py)
    through the emulated device kernels."""
    env = dict(os.environ, MSIM_LIB=emu_lib, HIPEMU_DIVERGENT="1")
    for args in ([os.path.join(ROOT, "tests", "test_checker_gpu.py"), "-k", "unique"], [os.path.join(ROOT, "tests", "test_rw_check_gpu.py")],
                 [os.path.join(ROOT, "tests", "test_lin_check_gpu.py")],   # the linearizability search beyond 64 configurations: LDS pools, the host for the rest
                 [os.path.join(ROOT, "tests", "test_checker_reference_vectors.py")],   # set-full / linearizability against the runs the reference docs print
                 [os.path.join(ROOT, "tests", "test_set_full_synthetic_gpu.py")]):   # check_kernel on synthetic histories: overtaking reads, failures, > 1024 elements, the slab's end
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
