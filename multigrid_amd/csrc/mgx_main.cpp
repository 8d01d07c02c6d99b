// mgx_main.cpp -- command-line driver with the reference's process contract
// (main.cpp:360-459): reads Input.mat from the working directory (variables
// `f`, optional `f_loc` = [x0 y0 dx dy]), runs Nv V-cycles on the GPU through
// the C ABI and writes Output.mat (`Sol`, `ResNorms`, `Times`).
//
// Flags and defaults are the reference's (boost::program_options,
// main.cpp:363-370): -v/--Nv (also the first positional argument, default
// 100), -b/--Nbefore (3), -a/--Nafter (3), -o/--omega (0.6), -c/--coarsest (2),
// -h/--help.  Added: --input / --output file names, --stepwise (one kernel per
// reference operator instead of the fused schedule), --exact-coarse (solve a
// coarsest grid other than 3x3 exactly instead of leaving it as it is),
// --ngpus N (split the grid into N j-slabs, one per device), --tail-m M (largest
// level every slab holds whole).
#include "../../include/mgx.h"
#include "mat4.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

void usage()
{
    printf("Allowed options:\n"
           "  -h [ --help ]                 produce help message\n"
           "  -v [ --Nv ] arg (=100)        number of V cycles to sweep\n"
           "  -b [ --Nbefore ] arg (=3)     number of Jacobi sweeps before refinement\n"
           "  -a [ --Nafter ] arg (=3)      number of Jacobi sweeps after refinement\n"
           "  -o [ --omega ] arg (=0.6)     relaxation parameter\n"
           "  -c [ --coarsest ] arg (=2)    threshold dimension to use a direct solver\n"
           "  --input arg (=Input.mat)      MAT-v4 file holding f and f_loc\n"
           "  --output arg (=Output.mat)    MAT-v4 file receiving Sol, ResNorms, Times\n"
           "  --stepwise                    one GPU kernel per reference operator\n"
           "  --exact-coarse                solve a coarsest grid other than 3x3 exactly\n"
           "  --ngpus arg (=1)              split the grid into this many j-slabs, one per GPU\n"
           "  --tail-m arg (=1025)          largest level every slab holds whole\n\n");
}

struct Args {
    int Nv = 100, Ndown = 3, Nup = 3, coarsest = 2, ngpus = 0, tail_m = 0;
    double omega = 0.6;
    std::string input = "Input.mat", output = "Output.mat";
    bool help = false, stepwise = false, exact_coarse = false;
};

bool parse_int(const char *s, int &out)
{
    char *end = nullptr;
    long v = strtol(s, &end, 10);
    if (!s[0] || *end) return false;
    out = (int)v;
    return true;
}
bool parse_double(const char *s, double &out)
{
    char *end = nullptr;
    double v = strtod(s, &end);
    if (!s[0] || *end) return false;
    out = v;
    return true;
}

// returns 0 ok, 2 on a malformed command line
int parse(int argc, const char *argv[], Args &a)
{
    bool positional_used = false;
    for (int k = 1; k < argc; k++) {
        std::string t = argv[k];
        std::string key, val;
        bool has_val = false;
        if (t.rfind("--", 0) == 0) {
            const size_t eq = t.find('=');
            key = t.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            if (eq != std::string::npos) { val = t.substr(eq + 1); has_val = true; }
        } else if (t.size() >= 2 && t[0] == '-' && !(t[1] >= '0' && t[1] <= '9') && t[1] != '.') {
            static const struct { char s; const char *l; } shorts[] = {
                {'h', "help"}, {'v', "Nv"}, {'b', "Nbefore"}, {'a', "Nafter"}, {'o', "omega"}, {'c', "coarsest"}};
            for (auto &m : shorts)
                if (m.s == t[1]) key = m.l;
            if (key.empty()) { fprintf(stderr, "unrecognised option '%s'\n", t.c_str()); return 2; }
            if (t.size() > 2) { val = t.substr(2); has_val = true; }  // -v30
        } else {
            if (positional_used) { fprintf(stderr, "too many positional options have been specified on the command line\n"); return 2; }
            positional_used = true;
            if (!parse_int(t.c_str(), a.Nv)) { fprintf(stderr, "the argument ('%s') for option '--Nv' is invalid\n", t.c_str()); return 2; }
            continue;
        }
        if (key == "help") { a.help = true; continue; }
        if (key == "stepwise") { a.stepwise = true; continue; }
        if (key == "exact-coarse") { a.exact_coarse = true; continue; }
        if (!has_val) {
            if (k + 1 >= argc) { fprintf(stderr, "the required argument for option '--%s' is missing\n", key.c_str()); return 2; }
            val = argv[++k];
        }
        bool ok = true;
        if (key == "Nv") ok = parse_int(val.c_str(), a.Nv);
        else if (key == "Nbefore") ok = parse_int(val.c_str(), a.Ndown);
        else if (key == "Nafter") ok = parse_int(val.c_str(), a.Nup);
        else if (key == "coarsest") ok = parse_int(val.c_str(), a.coarsest);
        else if (key == "omega") ok = parse_double(val.c_str(), a.omega);
        else if (key == "ngpus") ok = parse_int(val.c_str(), a.ngpus);
        else if (key == "tail-m") ok = parse_int(val.c_str(), a.tail_m);
        else if (key == "input") a.input = val;
        else if (key == "output") a.output = val;
        else { fprintf(stderr, "unrecognised option '--%s'\n", key.c_str()); return 2; }
        if (!ok) { fprintf(stderr, "the argument ('%s') for option '--%s' is invalid\n", val.c_str(), key.c_str()); return 2; }
    }
    return 0;
}

} // namespace

int main(int argc, const char *argv[])
{
    Args a;
    int rc = parse(argc, argv, a);
    if (rc) return rc;
    if (a.help) {  // main.cpp:379-383
        usage();
        return 0;
    }

    mat4::Reader in(a.input);  // main.cpp:393-396
    if (!in.is_open()) {
        printf("Error: cannot open %s\n", a.input.c_str());
        return 1;
    }
    if (!in.error().empty()) printf("Reading In File Content: %s\n", in.error().c_str());
    std::vector<double> f;
    int M = 0, N = 0;
    if (!in.read_double("f", f, M, N) || f.empty()) {
        printf("Error: %s holds no variable f\n", a.input.c_str());
        return 1;
    }
    double dx = 1.0;  // no f_loc: unit grid at the origin (DTSource/DTMesh2D.cpp:466-480)
    if (in.contains("f_loc")) {
        std::vector<double> loc;
        int lm = 0, ln = 0;
        if (!in.read_double("f_loc", loc, lm, ln) || loc.size() != 4) {  // DTSource/DTMesh2DGrid.cpp:279-292
            printf("ReadFromArray(DTMesh2DGrid): Incorrect size array.\n");
        } else {
            dx = loc[2];
        }
    }
    if (in.contains("f_dom")) {
        printf("Error: masked meshes (f_dom) are not supported\n");
        return 1;
    }
    if (M != N) {  // main.cpp:408-412
        printf("Error: Input is not square matrix!\n");
        return 1;
    }
    if (M % 2 == 0) {  // main.cpp:413-417
        printf("Error: Input has even rows and columns!\n");
        return 1;
    }
    if (a.Nv < 0 || a.Ndown < 0 || a.Nup < 0) {
        printf("Error: negative cycle or sweep count\n");
        return 1;
    }

    // main.cpp:420-439: zero initial guess, boundary_func == 0 on the boundary
    std::vector<double> u((size_t)M * (size_t)N, 0.0);
    std::vector<double> resnorm((size_t)a.Nv + 1, 0.0), times((size_t)a.Nv + 1, 0.0);

    mgx_options opt;
    memset(&opt, 0, sizeof(opt));
    opt.struct_size = (int)sizeof(opt);
    opt.device = -1;
    opt.verbose = 1;  // the reference prints "Direct solver not defined ..." (main.cpp:167)
    opt.schedule = a.stepwise ? MGX_SCHEDULE_STEPWISE : MGX_SCHEDULE_DEFAULT;
    opt.coarse_solve = a.exact_coarse ? MGX_COARSE_EXACT : MGX_COARSE_REFERENCE;
    opt.ngpus = a.ngpus;
    opt.tail_m = a.tail_m;
    opt.flags = MGX_FLAG_ZERO_GUESS;  // u is zero apart from its boundary (main.cpp:420-439)
    rc = mgx_solve2d(f.data(), u.data(), M, N, dx, a.Nv, a.Ndown, a.Nup, a.omega, a.coarsest, 0, resnorm.data(),
                     times.data(), &opt);  // main.cpp:445
    if (rc != MGX_OK) {
        printf("Error: %s\n", mgx_last_error());
        return 1;
    }

    mat4::Writer out(a.output);  // main.cpp:452-455
    if (!out.is_open() || !out.save(u.data(), M, N, "Sol") || !out.save(resnorm.data(), a.Nv + 1, 1, "ResNorms") ||
        !out.save(times.data(), a.Nv + 1, 1, "Times")) {
        printf("Error: cannot write %s\n", a.output.c_str());
        return 1;
    }
    return 0;
}
