// host/inclxpnd.cpp — recursive #include "..." expander (role of the reference's util/inclxpnd,
// /root/reference/util/inclxpnd/src/inclxpnd.cpp:8-41: flatten an app header and everything it includes into one
// stream, e.g. to paste it into Shadertoy).  Written from the tool's description, not from its source:
//   inclxpnd <file>            prints <file> with every   #include "x"   line replaced by the expansion of x
// (paths resolve relative to the including file; <angle> includes and unreadable files are kept as they are; a file
// already being expanded is not entered again, so include cycles terminate).
#include <cstdio>
#include <fstream>
#include <iostream>
#include <set>
#include <string>

static std::string dir_of(const std::string& path) {
    const size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? std::string() : path.substr(0, p + 1);
}

static bool expand(const std::string& path, std::set<std::string>& open, std::ostream& out) {
    std::ifstream in(path);
    if (!in) return false;
    if (!open.insert(path).second) return true;          // cycle: already being expanded
    std::string line;
    while (std::getline(in, line)) {
        size_t i = line.find_first_not_of(" \t");
        bool done = false;
        if (i != std::string::npos && line[i] == '#') {
            size_t j = line.find_first_not_of(" \t", i + 1);
            if (j != std::string::npos && line.compare(j, 7, "include") == 0) {
                const size_t q0 = line.find('"', j + 7);
                const size_t q1 = q0 == std::string::npos ? q0 : line.find('"', q0 + 1);
                if (q1 != std::string::npos) done = expand(dir_of(path) + line.substr(q0 + 1, q1 - q0 - 1), open, out);
            }
        }
        if (!done) out << line << '\n';
    }
    open.erase(path);
    return true;
}

int main(int argc, char** argv) {
    if (argc != 2) { std::fprintf(stderr, "usage: inclxpnd <file>\n"); return 2; }
    std::set<std::string> open;
    if (!expand(argv[1], open, std::cout)) { std::fprintf(stderr, "inclxpnd: cannot read %s\n", argv[1]); return 1; }
    return 0;
}
