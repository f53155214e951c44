// rb_build -- the reference's index builder CLI (reference src/rb_build.cpp) for this engine:
//   rb_build [-o out_prefix] [-s] [-m] [-l] [-f [-k K]] [-a] <input_prefix>
// reads <input_prefix>.bwt (+ .ssa/.esa with -s, + .docs with -l) exactly as the reference's
// constructors do (rle_string.hpp:44-97, toehold_sa.hpp:27-35,133-155) and writes the engine's
// native cache <out_prefix>.rbgpu, which rb_align / rb_markers / rbg_load pick up when no .rbwt is
// there.  -f / -a write the reference's own text <out_prefix>.ftab (rowbowt.hpp:726-744,
// ftab.hpp:29-34), computed on the GPU.
//
// Differences, all reported on stderr:
//   -m   the reference parses the raw <input_prefix>.ma with pfbwt-f's MarkerArray constructor, which
//        is not vendored; here -m folds an already serialised <input_prefix>.mab into the cache.
//   -x   fbb_string indexes are not supported.
//   The sdsl-serialised .rbwt/.tsa are not written (this engine does not need them; an existing
//   reference-built index converts with `rb_build --from-index <prefix>`).
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "../../include/rbg.h"

namespace {

struct Args {  // RowBowtConstructArgs, rowbowt_io.hpp:30-47
    std::string prefix, inpre;
    int ma = 0, tsa = 0, dl = 0, ft = 0, ft_only = 0, fbb = 0, from_index = 0, device = 0;
    uint64_t k = 10;
};

void print_help() {  // rb_build.cpp:11-20
    fprintf(stderr, "rb_build\n");
    fprintf(stderr, "Usage: rb_build [options] <index_prefix>\n");
    fprintf(stderr, "    --output-prefix/-o <basename>    output prefix\n");
    fprintf(stderr, "    --tsa/-s                         build toehold suffix array (<pre>.ssa, <pre>.esa)\n");
    fprintf(stderr, "    --ma/-m                          include the marker array (<pre>.mab)\n");
    fprintf(stderr, "    --dl/-l                          include the document list (<pre>.docs)\n");
    fprintf(stderr, "    --ft/-f [-k <int>]               construct offset table <out>.ftab\n");
    fprintf(stderr, "    --ftab-only/-a                   only construct the offset table from an existing index\n");
    fprintf(stderr, "    --from-index                     convert <pre>.rbwt/.tsa/.mab/.docs instead of raw inputs\n");
    fprintf(stderr, "    --gpu <n>                        HIP device ordinal for -f/-a (default 0)\n");
    fprintf(stderr, "    <input_prefix>                   index prefix\n");
}

Args parse_args(int argc, char **argv) {  // rb_build.cpp:22-95
    static Args args;
    static struct option long_options[] = {{"output-prefix", required_argument, 0, 'o'},
                                           {"tsa", no_argument, 0, 's'},
                                           {"dl", no_argument, 0, 'l'},
                                           {"ftab-only", no_argument, 0, 'a'},
                                           {"ma", no_argument, 0, 'm'},
                                           {"ft", no_argument, 0, 'f'},
                                           {"fbb", no_argument, 0, 'x'},
                                           {"from-index", no_argument, &args.from_index, 1},
                                           {"gpu", required_argument, 0, 'G'},
                                           {0, 0, 0, 0}};
    int c, long_index = 0;
    while ((c = getopt_long(argc, argv, "xo:k:lfsmha", long_options, &long_index)) != -1) {
        switch (c) {
            case 0: break;
            case 'x': args.fbb = 1; break;
            case 'o': args.prefix = optarg; break;
            case 's': args.tsa = 1; break;
            case 'm': args.ma = 1; break;
            case 'l': args.dl = 1; break;
            case 'f': args.ft = 1; break;
            case 'a': args.ft_only = 1; break;
            case 'k': args.k = std::strtoull(optarg, nullptr, 10); break;
            case 'G': args.device = atoi(optarg); break;
            case 'h': print_help(); exit(0);
            case '?': break;  // the reference ignores unknown options (rb_build.cpp:64-65)
            default: print_help(); exit(1);
        }
    }
    if (argc - optind < 1) {
        fprintf(stderr, "no argument provided\n");
        exit(1);
    }
    args.inpre = argv[optind++];
    if (args.prefix.empty()) args.prefix = args.inpre;
    return args;
}

bool file_exists(const std::string &f) { return std::ifstream(f).good(); }

[[noreturn]] void die(const std::string &what, int rc) {
    std::cerr << "rb_build: " << what << ": " << rbg_strerror(rc) << std::endl;
    exit(1);
}

void file_ne_error(const std::string &f) {  // rowbowt_io.hpp:23-26
    std::cerr << f << " does not exist" << std::endl;
    exit(1);
}

void write_ftab(const Args &args) {
    rbg_index *ix = nullptr;
    int rc = rbg_load(args.prefix.c_str(), RBG_LOAD_NONE, args.device, &ix);  // .rbwt, else .rbgpu (rowbowt_io.hpp:127-139)
    if (rc == RBG_EIO) rc = rbg_build_from_files((args.inpre + ".bwt").c_str(), nullptr, nullptr, args.device, &ix);
    else if (!rc) std::cerr << "loading rbwt file" << std::endl;
    if (rc) die("loading the index for the ftab", rc);
    if ((rc = rbg_write_ftab(ix, args.k, (args.prefix + ".ftab").c_str()))) die("writing " + args.prefix + ".ftab", rc);
    rbg_free(ix);
}

}  // namespace

int main(int argc, char **argv) {
    const Args args = parse_args(argc, argv);
    if (args.fbb) {
        std::cerr << "rb_build: fbb_string indexes are not supported by this engine" << std::endl;
        return 1;
    }
    if (args.ft_only) {  // rb_build.cpp:108-109
        write_ftab(args);
        return 0;
    }
    const std::string out = args.prefix + ".rbgpu";
    int rc;
    if (args.from_index) {
        int flags = RBG_LOAD_NONE;
        if (args.tsa) flags |= RBG_LOAD_SA;
        if (args.ma) flags |= RBG_LOAD_MA;
        if (args.dl) flags |= RBG_LOAD_DL;
        std::cerr << "converting " << args.inpre << ".rbwt to " << out << std::endl;
        if ((rc = rbg_convert_index(args.inpre.c_str(), flags, out.c_str()))) die("converting " + args.inpre, rc);
    } else {
        std::cerr << "constructing using rle_string (flat run-length BWT)" << std::endl;  // rowbowt_io.hpp:52
        const std::string bwt = args.inpre + ".bwt", ssa = args.inpre + ".ssa", esa = args.inpre + ".esa";
        const std::string mab = args.inpre + ".mab", docs = args.inpre + ".docs";
        if (!file_exists(bwt)) file_ne_error(bwt);
        if (args.tsa) {  // rowbowt_io.hpp:65-67
            if (!file_exists(ssa)) file_ne_error(ssa);
            if (!file_exists(esa)) file_ne_error(esa);
        }
        if (args.ma && !file_exists(mab)) {
            std::cerr << "rb_build: -m needs a serialised marker array " << mab
                      << " (the raw .ma reader belongs to pfbwt-f, which is not part of this engine)" << std::endl;
            return 1;
        }
        rc = rbg_convert_raw(bwt.c_str(), args.tsa ? ssa.c_str() : nullptr, args.tsa ? esa.c_str() : nullptr,
                             args.ma ? mab.c_str() : nullptr, (args.dl && file_exists(docs)) ? docs.c_str() : nullptr, out.c_str());
        if (rc) die("building " + out, rc);
        if (args.dl && docs != args.prefix + ".docs") {  // rowbowt_io.hpp:73-80: the .docs file is copied
            std::ifstream ifs(docs);
            std::ofstream ofs(args.prefix + ".docs");
            ofs << ifs.rdbuf();
        }
    }
    if (args.ft) write_ftab(args);  // rowbowt_io.hpp:82-88
    return 0;
}
