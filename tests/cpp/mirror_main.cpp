// Drives libshk_hip.so through the header-only C++ mirror of the reference's AssemblyHelper
// (include/sparrowhawk_asm.hpp), in the order www/src/workers/Assembler.ts:73-139 uses.
// usage: mirror_main <fastq> <k> <min_count> <out_prefix>     exit 3 = helper could not be created
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>
#include "sparrowhawk_asm.hpp"

static std::vector<std::string> g_states;
static void on_state(const char *s, void *) { g_states.push_back(s); }

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> fq((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    try {
        auto h = sparrowhawk::AssemblyHelper::new_((uint32_t)atoi(argv[2]), true, (uint32_t)atoi(argv[3]), 20, 0, false, false,
                                                    false, false);
        h.on_state(on_state, nullptr);
        h.preprocess(fq);
        const std::string pre = h.get_preprocessing_info();
        h.assemble();
        const std::string as = h.get_assembly();
        std::ofstream(std::string(argv[4]) + ".pre.json") << pre;
        std::ofstream(std::string(argv[4]) + ".asm.json") << as;
        std::ofstream st(std::string(argv[4]) + ".states");
        for (auto &s : g_states) st << s << "\n";
    } catch (const std::exception &e) {
        fprintf(stderr, "AssemblyHelper: %s\n", e.what());
        return 3;
    }
    return 0;
}
