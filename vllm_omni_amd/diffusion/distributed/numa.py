"""Pin a GPU worker's host threads to the NUMA node its GPU hangs off.

One process per GPU (reference WorkerProc, vllm_omni/diffusion/worker/gpu_worker.py:143-314): on a two-socket MI355X node four
GPUs sit behind each socket, and a worker whose launch thread runs on the far socket pays a cross-socket hop for every
kernel-launch doorbell, pinned-buffer copy and queue wake-up.  The reference leaves placement to the OS.  Here each rank
restricts itself to the CPUs of `/sys/bus/pci/devices/<bdf>/numa_node` of its device (Linux sysfs; anything missing — a
container without sysfs, a single-node host, numa_node = -1 — leaves the affinity untouched and says so)."""
from __future__ import annotations

import os


def parse_cpulist(text: str) -> list[int]:
    """'0-15,32-47' -> [0..15, 32..47] (the kernel's cpulist format)."""
    cpus: list[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus += list(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return cpus


def gpu_pci_address(device_index: int) -> str | None:
    try:
        import torch

        p = torch.cuda.get_device_properties(device_index)
        return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:  # noqa: BLE001 — no GPU / an older torch without the pci_* fields
        return None


def numa_cpus_of(pci_address: str, sysfs: str = "/sys") -> tuple[int, list[int]] | None:
    """(node, cpus of that node) of a PCI device, or None when sysfs does not say."""
    try:
        with open(os.path.join(sysfs, "bus/pci/devices", pci_address, "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(os.path.join(sysfs, "devices/system/node", f"node{node}", "cpulist")) as f:
            cpus = parse_cpulist(f.read())
        return (node, cpus) if cpus else None
    except (OSError, ValueError):
        return None


def pin_to_gpu_numa(device_index: int, sysfs: str = "/sys") -> dict:
    """Restrict this process to the CPUs of its GPU's NUMA node (intersected with what the cgroup already allows).  Returns
    {"pinned": bool, "node": n | None, "cpus": count, "reason": str}."""
    addr = gpu_pci_address(device_index)
    if addr is None:
        return {"pinned": False, "node": None, "cpus": 0, "reason": "no PCI address for the device"}
    got = numa_cpus_of(addr, sysfs)
    if got is None:
        return {"pinned": False, "node": None, "cpus": 0, "reason": f"sysfs has no NUMA node for {addr}"}
    node, cpus = got
    try:
        allowed = os.sched_getaffinity(0)
        want = sorted(set(cpus) & allowed)
        if not want:
            return {"pinned": False, "node": node, "cpus": 0, "reason": "the node's CPUs are outside this process's cpuset"}
        if set(want) == set(allowed):
            return {"pinned": False, "node": node, "cpus": len(want), "reason": "already confined to that node"}
        os.sched_setaffinity(0, want)
        return {"pinned": True, "node": node, "cpus": len(want), "reason": f"{addr} -> node {node}"}
    except (AttributeError, OSError) as e:
        return {"pinned": False, "node": node, "cpus": 0, "reason": f"sched_setaffinity: {e}"}
