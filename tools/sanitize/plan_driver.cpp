#include <fstream>
#include <iostream>
#include <string>
#include "sqlfront.h"
int main(int argc, char** argv) {
    std::ifstream sf(argv[1]);
    std::vector<rsq::Table*> db;
    std::string kw;
    while (sf >> kw) {
        rsq::Table* t = new rsq::Table();          // never deleted: ~Table lives in the HIP runtime part of the library
        int nc; sf >> t->name >> t->nRows >> nc;
        for (int i = 0; i < nc; i++) { rsq::TableColumn c; sf >> c.name >> c.type.tag >> c.type.precision >> c.type.scale >> c.type.len; t->cols.push_back(c); }
        db.push_back(t);
    }
    std::ifstream f(argv[2]);
    std::string line; long ok = 0, err = 0; size_t bytes = 0;
    while (std::getline(f, line)) {
        try {
            rsq::ExprPool pool; rsq::sql::Statement st; rsq::sql::PlanDesc pd;
            rsq::sql::parse(line, pool, st);
            rsq::sql::planSelect(st, pool, db, pd);
            bytes += rsq::sql::dumpPlan(pd.desc, db).size();
            ok++;
        } catch (const rsq::Error&) { err++; }
    }
    std::cout << "planned " << ok << " refused " << err << " dump bytes " << bytes << std::endl;
    return 0;
}
