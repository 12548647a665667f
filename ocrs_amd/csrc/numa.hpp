// NUMA placement of an engine-group member's host side (no HIP in this header; host-only tests drive it through the ABI).
//
// SURVEY.md §8e names the host as the expected scaling limiter of the 8-GPU stream: a member's worker thread should run on —
// and its page-locked staging be first touched from — the host NUMA node its GPU hangs off.  The node comes from sysfs
// (/sys/bus/pci/devices/<bus id>/numa_node, the bus id from hipDeviceGetPCIBusId), the node's CPUs from
// /sys/devices/system/node/node<N>/cpulist.  Everything falls back silently: no sysfs, node -1 (single-node hosts,
// containers), an empty list or a refused sched_setaffinity leave the thread where it was.
#pragma once
#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace ocrs {
namespace numa {

// "0-3,8,10-11" -> cpu numbers; false on a malformed list
inline bool parse_cpulist(const char* s, std::vector<int>* cpus) {
    cpus->clear();
    if (!s) return false;
    const char* p = s;
    while (*p == ' ' || *p == '\n' || *p == '\t') p++;
    if (!*p) return true;   // empty list
    for (;;) {
        char* end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p || a < 0 || a > 65535) return false;
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            if (end == p + 1 || b < a || b > 65535) return false;
            p = end;
        }
        for (long c = a; c <= b; c++) cpus->push_back((int)c);
        while (*p == ' ' || *p == '\n' || *p == '\t') p++;
        if (!*p) return true;
        if (*p != ',') return false;
        p++;
    }
}

inline bool read_file(const std::string& path, std::string* out) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    *out = buf;
    return true;
}

// NUMA node of a PCI device ("0000:c1:00.0", any case); -1 if unknown
inline int node_of_pci(const char* bus_id, const char* sysfs_root = "/sys") {
    if (!bus_id || !*bus_id) return -1;
    std::string id(bus_id);
    for (char& c : id) c = (char)tolower((unsigned char)c);
    std::string txt;
    if (!read_file(std::string(sysfs_root) + "/bus/pci/devices/" + id + "/numa_node", &txt)) return -1;
    char* end = nullptr;
    const long v = strtol(txt.c_str(), &end, 10);
    return end == txt.c_str() ? -1 : (int)v;
}

inline bool cpus_of_node(int node, std::vector<int>* cpus, const char* sysfs_root = "/sys") {
    cpus->clear();
    if (node < 0) return false;
    std::string txt;
    if (!read_file(std::string(sysfs_root) + "/devices/system/node/node" + std::to_string(node) + "/cpulist", &txt)) return false;
    return parse_cpulist(txt.c_str(), cpus) && !cpus->empty();
}

// Binds the calling thread to `cpus` for the lifetime of the object and puts the previous mask back afterwards.
class BindScope {
  public:
    explicit BindScope(const std::vector<int>& cpus) {
        if (cpus.empty()) return;
        if (sched_getaffinity(0, sizeof old_, &old_) != 0) return;
        cpu_set_t want;
        CPU_ZERO(&want);
        int n = 0;
        for (int c : cpus)
            if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &old_)) { CPU_SET(c, &want); n++; }   // never widen what the caller was given
        if (n == 0) return;
        bound_ = sched_setaffinity(0, sizeof want, &want) == 0;
    }
    ~BindScope() {
        if (bound_) (void)sched_setaffinity(0, sizeof old_, &old_);
    }
    BindScope(const BindScope&) = delete;
    BindScope& operator=(const BindScope&) = delete;
    bool bound() const { return bound_; }

  private:
    cpu_set_t old_;
    bool bound_ = false;
};

inline int affinity_count() {
    cpu_set_t s;
    if (sched_getaffinity(0, sizeof s, &s) != 0) return -1;
    return CPU_COUNT(&s);
}

}  // namespace numa
}  // namespace ocrs
