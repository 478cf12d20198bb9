#include "hostbind.h"

#include <sched.h>

#include <cstdio>
#include <cstdlib>

namespace reve {

std::vector<int> parse_cpulist(const std::string& s)
{
    std::vector<int> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && (s[i] == ',' || s[i] == ' ' || s[i] == '\n' || s[i] == '\t')) ++i;
        if (i >= s.size() || s[i] < '0' || s[i] > '9') break;
        long a = 0, b;
        while (i < s.size() && s[i] >= '0' && s[i] <= '9' && a < 100000) a = a * 10 + (s[i++] - '0');
        b = a;
        if (i < s.size() && s[i] == '-') {
            ++i;
            b = 0;
            if (i >= s.size() || s[i] < '0' || s[i] > '9') break;
            while (i < s.size() && s[i] >= '0' && s[i] <= '9' && b < 100000) b = b * 10 + (s[i++] - '0');
        }
        if (a > b || b >= CPU_SETSIZE) break;
        for (long c = a; c <= b; ++c) out.push_back((int)c);
    }
    return out;
}

static std::string slurp(const std::string& path)
{
    std::string s;
    if (FILE* f = std::fopen(path.c_str(), "r")) {
        char buf[4096];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) s.append(buf, n);
        std::fclose(f);
    }
    while (!s.empty() && (s.back() == '\n' || s.back() == ' ')) s.pop_back();
    return s;
}

static std::string lower(std::string s)
{
    for (char& c : s)
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
    return s;
}

std::string pci_local_cpulist(const std::string& bus_id, const std::string& root)
{
    if (bus_id.empty()) return "";
    return slurp(root + "/sys/bus/pci/devices/" + lower(bus_id) + "/local_cpulist");
}

int pci_numa_node(const std::string& bus_id, const std::string& root)
{
    if (bus_id.empty()) return -1;
    const std::string s = slurp(root + "/sys/bus/pci/devices/" + lower(bus_id) + "/numa_node");
    if (s.empty()) return -1;
    return std::atoi(s.c_str());
}

int bind_this_thread(const std::string& cpulist)
{
    const std::vector<int> want = parse_cpulist(cpulist);
    if (want.empty()) return 0;
    cpu_set_t cur, next;
    if (sched_getaffinity(0, sizeof(cur), &cur) != 0) return 0;
    CPU_ZERO(&next);
    int n = 0;
    for (int c : want)
        if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &cur)) { CPU_SET(c, &next); ++n; }
    if (n == 0) return 0;                    // nothing in common with what this process may use: leave the thread alone
    if (sched_setaffinity(0, sizeof(next), &next) != 0) return 0;
    return n;
}

}  // namespace reve
