// Drives the REFERENCE's PerfStats (kfusion/thirdparty/perfstats.h, included as it lies; standard headers only) the way the GUI mains
// do (mainQt.cpp:70-87 storeStats, :176 print_all_data): samples read from a text file - first line `name:TYPE ...`, then one row of
// numbers per frame - and print_all_data to stdout (`-`) or to a file, the closing "# name Mean: ..." block always on stdout.
// Test infrastructure: its output on seeded inputs is tests/golden/perfstats_vectors.json (tests/golden/make_perfstats_vectors.py).
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <perfstats.h>

static PerfStats::Type type_of(const std::string &t) {
    if (t == "TIME") return PerfStats::TIME;
    if (t == "DISTANCE") return PerfStats::DISTANCE;
    if (t == "INT") return PerfStats::INT;
    if (t == "FRAME") return PerfStats::FRAME;
    return PerfStats::DOUBLE;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    std::ifstream in(argv[1]);
    std::string line;
    std::getline(in, line);
    std::vector<std::string> names;
    std::vector<PerfStats::Type> types;
    {
        std::istringstream h(line);
        std::string tok;
        while (h >> tok) {
            const size_t c = tok.find(':');
            names.push_back(tok.substr(0, c));
            types.push_back(type_of(tok.substr(c + 1)));
        }
    }
    PerfStats stats;
    while (std::getline(in, line)) {
        std::istringstream r(line);
        double v;
        for (size_t i = 0; i < names.size() && (r >> v); ++i) stats.sample(names[i], v, types[i]);
    }
    if (std::string(argv[2]) == "-") {
        stats.print_all_data(std::cout);
    } else {
        std::ofstream out(argv[2]);
        stats.print_all_data(out);
    }
    return 0;
}
