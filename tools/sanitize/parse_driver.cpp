#include <fstream>
#include <iostream>
#include <string>
#include "sqlfront.h"
int main(int argc, char** argv) {
    std::ifstream f(argv[1]);
    std::string line; long ok = 0, err = 0; size_t bytes = 0;
    while (std::getline(f, line)) {
        for (auto& c : line) if (c == '\x01') c = '\n';
        try {
            rsq::ExprPool pool; rsq::sql::Statement st;
            rsq::sql::parse(line, pool, st);
            bytes += rsq::sql::dumpStatement(st).size();
            ok++;
        } catch (const rsq::Error&) { err++; }
    }
    std::cout << "parsed " << ok << " refused " << err << " dump bytes " << bytes << std::endl;
    return 0;
}
