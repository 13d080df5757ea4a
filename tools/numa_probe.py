#!/usr/bin/env python3
"""GPU box: does the NUMA node of a page-locked host window matter to the zero-copy path?  The window's pages are first
touched (and so placed) by a thread pinned to each node in turn, registered, and handed to rr_block_work: us per call for a
block with wide coalesced reads (FftFilter) and for the RTL-SDR byte chain (2-byte elements, 12-byte lane stride).
Round 4 (GPU on node 1 of 2): no difference beyond noise — FmChainU8 122-124 us on either node, FftFilter 153-171; the one
outlier per process (the SECOND block instance's first passes, 330-470 us) follows the order of the measurements, not the node."""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustradio_amd as rr

def cpus_of(node):
    out = []
    for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out

nodes = sorted(int(p.rsplit("node", 1)[1]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
for p in glob.glob("/sys/class/drm/card*/device/numa_node") + glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
    try:
        txt = open(p).read()
        if "numa_node" in p: print(p, txt.strip())
    except OSError:
        pass
print("nodes:", nodes, "allowed cpus:", len(os.sched_getaffinity(0)))
try:
    import torch
    bus = torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else None
    print("HIP device 0 pci bus id:", bus)
    for p in glob.glob("/sys/bus/pci/devices/*/numa_node"):
        dev = p.split("/")[-2]
        if bus is not None and dev.split(":")[1].lower() == f"{bus:02x}":
            print("  ", dev, "numa_node", open(p).read().strip())
except Exception as e:
    print("no torch device info:", e)
allowed = os.sched_getaffinity(0)
t2 = rr.low_pass_complex(2.4e6, 100e3, 12.5e3)
t1 = rr.low_pass_complex(10e6, 1e6, 60e3)
rng = np.random.default_rng(1)
def timeit(blk, x, out, reps=200):
    for _ in range(5): blk.work_into(x, out, len(out))
    t0 = time.perf_counter()
    for _ in range(reps): blk.work_into(x, out, len(out))
    return (time.perf_counter() - t0) / reps * 1e6
for node in nodes * 3:
    cp = set(cpus_of(node)) & allowed
    if not cp:
        print(f"node {node}: no allowed cpus"); continue
    os.sched_setaffinity(0, cp)
    xb = rng.integers(0, 256, 4_096_000, dtype=np.uint8); ob = np.zeros(1_024_000, np.float32)
    xc = (rng.uniform(-1, 1, 512_000) + 1j * rng.uniform(-1, 1, 512_000)).astype(np.complex64); oc = np.zeros(512_000, np.complex64)
    for a in (xb, ob, xc, oc): rr.host_register(a)
    os.sched_setaffinity(0, allowed)
    print(f"windows first touched on node {node}: FmChainU8 {timeit(rr.FmChainU8(t2, 1, 6, 1.0), xb, ob):7.1f} us   FftFilter {timeit(rr.FftFilter(t1), xc, oc):7.1f} us", flush=True)
    for a in (xb, ob, xc, oc): rr.host_unregister(a)
