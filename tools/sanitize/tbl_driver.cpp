// ASan / UBSan driver for the '.tbl' ingest (resql_amd/csrc/tbl.cpp): parses a file with a given schema, with 1 and 8 threads.
// usage: tbl_driver FILE TERMINATOR TYPE...   (TYPE: INT BIGINT DATE CHAR:n VARCHAR:n DECIMAL:p:s)
#include <cstring>
#include <iostream>
#include "engine.h"
int main(int argc, char** argv) {
    std::vector<rsq::Type> types;
    for (int i = 3; i < argc; i++) {
        std::string t = argv[i];
        if (t == "INT") types.push_back(rsq::Type(RSQ_INT));
        else if (t == "BIGINT") types.push_back(rsq::Type(RSQ_BIGINT));
        else if (t == "DATE") types.push_back(rsq::Type(RSQ_DATE));
        else if (t.compare(0, 5, "CHAR:") == 0) { rsq::Type x(RSQ_CHAR); x.len = atoi(t.c_str() + 5); types.push_back(x); }
        else if (t.compare(0, 8, "VARCHAR:") == 0) { rsq::Type x(RSQ_VARCHAR); x.len = atoi(t.c_str() + 8); types.push_back(x); }
        else if (t.compare(0, 8, "DECIMAL:") == 0) { int p = 0, s = 0; sscanf(t.c_str() + 8, "%d:%d", &p, &s); types.push_back(rsq::Type::decimal(p, s)); }
    }
    for (int threads : {1, 8}) {
        try {
            std::vector<std::vector<uint8_t>> cols; int64_t n = 0;
            rsq::parseTblFile(argv[1], types, argv[2][0], threads, cols, n);
            size_t bytes = 0; for (auto& c : cols) bytes += c.size();
            std::cout << "threads " << threads << ": " << n << " rows, " << bytes << " column bytes" << std::endl;
        } catch (const rsq::Error& e) { std::cout << "threads " << threads << ": refused: " << e.what() << std::endl; }
    }
    return 0;
}
