"""CPU: the drivers' host placement (benchmarks/common/gab_driver.h) -- PCI bus id -> NUMA node -> cpu list, against a made-up
sysfs tree ($GAB_SYSFS_ROOT).  The reference pins its threads with OMP_PROC_BIND=true OMP_PLACES=cores
(bsw/scripts/regression_small.sh:52); here a GPU's worker threads go to the cores of the node the card hangs off."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = r"""
#include "benchmarks/common/gab_driver.h"
int main(int argc, char **argv) {
    cpu_set_t s;
    /* 1. cpu lists as the kernel prints them */
    printf("%d\n", gab_parse_cpulist("0-3,8,10-11\n", &s));
    printf("%d%d%d%d%d\n", CPU_ISSET(0, &s), CPU_ISSET(3, &s), CPU_ISSET(4, &s), CPU_ISSET(8, &s), CPU_ISSET(11, &s));
    printf("%d\n", gab_parse_cpulist("", &s));
    printf("%d\n", gab_parse_cpulist("5-2", &s));
    /* 2. bus id (any case) -> node -> cpus */
    for (int k = 1; k < argc; k++) {
        const int node = gab_numa_node_of_pci(argv[k]);
        printf("%s node %d cpus %d\n", argv[k], node, node >= 0 ? gab_node_cpus(node, &s) : -1);
    }
    return 0;
}
"""


def test_node_lookup(tmp_path):
    root = tmp_path / "fake"
    for busid, node in (("0000:c1:00.0", 1), ("0000:05:00.0", 0), ("0000:75:00.0", -1)):
        d = root / "sys" / "bus" / "pci" / "devices" / busid
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
    for node, cpus in ((0, "0-15,128-143"), (1, "16-31,144-159")):
        d = root / "sys" / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    src = tmp_path / "t.c"
    src.write_text(PROG)
    lib = os.path.join(ROOT, "genarchbench_amd")
    exe = str(tmp_path / "t")
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-I", ROOT, "-I", os.path.join(ROOT, "include"), str(src), "-o", exe,
                           "-L", lib, "-lgab_hip", f"-Wl,-rpath,{lib}", "-lpthread", "-ldl"])
    out = subprocess.run([exe, "0000:C1:00.0", "0000:05:00.0", "0000:75:00.0", "0000:ff:00.0"], capture_output=True, text=True,
                         env=dict(os.environ, GAB_SYSFS_ROOT=str(root)), timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert lines[:4] == ["7", "11011", "0", "-1"]
    assert lines[4:] == ["0000:C1:00.0 node 1 cpus 32", "0000:05:00.0 node 0 cpus 32", "0000:75:00.0 node -1 cpus -1",
                         "0000:ff:00.0 node -1 cpus -1"]
