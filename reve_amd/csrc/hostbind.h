// Where a GPU's host-side work should run: the CPUs (and NUMA node) next to its PCIe root, read from sysfs.
// "one host worker thread + 3 streams + pinned ring per GPU, pinned to the GPU's NUMA node" — the feeder
// and buffer-allocator threads of a GPU bind themselves to its local CPUs before their first pinned allocation, so that the
// pinned pages (first touch) and the thread that fills them sit on the node the GPU's DMA engines reach without crossing sockets.
// Pure sysfs + sched code: no HIP call in here (the engine supplies the PCI bus id), so it is part of the CPU sanitizer builds.
#pragma once
#include <string>
#include <vector>

namespace reve {
// "0-3,8,10-11" -> {0,1,2,3,8,10,11}; malformed input -> what could be parsed
std::vector<int> parse_cpulist(const std::string& s);
// local_cpulist / numa_node of the PCI device `bus_id` ("0000:c1:00.0"); `root` is prefixed to /sys (tests use a fake tree).
// Empty / -1 when the files do not exist (a box without NUMA information).
std::string pci_local_cpulist(const std::string& bus_id, const std::string& root = "");
int pci_numa_node(const std::string& bus_id, const std::string& root = "");
// Restricts the CALLING thread to the CPUs of `cpulist` that its current affinity mask allows (a container's mask is never
// widened).  Returns the number of CPUs the thread is bound to, 0 if nothing was changed (empty list or no CPU in common).
int bind_this_thread(const std::string& cpulist);
}  // namespace reve
