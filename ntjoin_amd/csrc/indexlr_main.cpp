// indexlr_main.cpp -- command-line twin of btllib's `indexlr` for the flags ntJoin uses, running on the GPU
// through the C-ABI of libntjoin_mx.so.  Replaces the recipe at reference ntJoin:204-205
//     indexlr --seq --long --pos -k $(k) -w $(w) -t $(t) $< > $@
// and run_indexlr()'s spelling at reference bin/ntjoin_utils.py:198 (`-k32 -w1000 -t4 ... -o file`).
// stdout carries only the TSV; diagnostics go to stderr; any failure exits non-zero (ntJoin runs its recipes
// under `bash -e -o pipefail`, reference ntJoin:89).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include <sys/stat.h>

#include "ntjoin_mx.h"

// a partial output is removed only when its name is a regular file: a FIFO, /dev/stdout or a process substitution is not ours to unlink
static void remove_partial(const char *path)
{
    struct stat sb;
    if (strcmp(path, "-") != 0 && lstat(path, &sb) == 0 && S_ISREG(sb.st_mode)) remove(path);
}

static void usage(FILE *f)
{
    fputs("Usage: indexlr -k K -w W [--id] [--pos] [--strand] [--seq] [--long] [-t T] [-o FILE] FASTA[.gz]\n"
          "  GPU (MI355X) minimizer sketcher, output-compatible with btllib indexlr for these options:\n"
          "  -k K           k-mer size (required)\n"
          "  -w W           window size in k-mers (required)\n"
          "  --id           print the record id as first column (default, always on)\n"
          "  --pos          print minimizer positions            (hash:pos)\n"
          "  --strand       print minimizer strands              (hash[:pos]:+|-)\n"
          "  --seq          print minimizer k-mer sequences      (hash[:pos][:strand]:seq)\n"
          "  --long         accepted for compatibility (records are sketched on the GPU regardless of length)\n"
          "  -t T           host worker threads for reading the FASTA / writing the TSV (the GPU does the hashing)\n"
          "  -o FILE        write to FILE instead of stdout\n"
          "  --variant v2|v1   canonical hash: v2 = fwd+rev (current btllib, default), v1 = min(fwd,rev)\n"
          "  --device N     HIP device ordinal (default: current)\n"
          "  --dense        disable the sparse-candidate fast path\n"
          "  -v             verbose (statistics on stderr)\n",
          f);
}

static bool opt_val(int argc, char **argv, int &i, const char *name, const char **val)
{
    size_t n = strlen(name);
    if (strncmp(argv[i], name, n) != 0) return false;
    if (argv[i][n] == 0) {
        if (i + 1 >= argc) return false;
        *val = argv[++i];
        return true;
    }
    if (name[1] != '-') {  // short option glued to its value: -k32
        *val = argv[i] + n;
        return true;
    }
    if (argv[i][n] == '=') {
        *val = argv[i] + n + 1;
        return true;
    }
    return false;
}

int main(int argc, char **argv)
{
    unsigned k = 0, w = 0, threads = 0;
    int with_pos = 0, with_strand = 0, with_seq = 0, verbose = 0, device = -1, dense = 0;
    unsigned variant = MXG_VARIANT_V2_SUM;
    const char *out = "-", *in = nullptr, *v = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--help") || !strcmp(argv[i], "-h")) { usage(stdout); return 0; }
        else if (!strcmp(argv[i], "--pos")) with_pos = 1;
        else if (!strcmp(argv[i], "--strand")) with_strand = 1;
        else if (!strcmp(argv[i], "--seq")) with_seq = 1;
        else if (!strcmp(argv[i], "--long") || !strcmp(argv[i], "--id")) { /* no-op */ }
        else if (!strcmp(argv[i], "--dense")) dense = 1;
        else if (!strcmp(argv[i], "-v")) verbose = 1;
        else if (opt_val(argc, argv, i, "--variant", &v)) variant = (!strcmp(v, "v1") || !strcmp(v, "min")) ? MXG_VARIANT_V1_MIN : MXG_VARIANT_V2_SUM;
        else if (opt_val(argc, argv, i, "--device", &v)) device = atoi(v);
        else if (opt_val(argc, argv, i, "-k", &v)) k = (unsigned)strtoul(v, nullptr, 10);
        else if (opt_val(argc, argv, i, "-w", &v)) w = (unsigned)strtoul(v, nullptr, 10);
        else if (opt_val(argc, argv, i, "-t", &v)) threads = (unsigned)strtoul(v, nullptr, 10);
        else if (opt_val(argc, argv, i, "-o", &v)) out = v;
        else if (argv[i][0] == '-' && argv[i][1] != 0) {
            fprintf(stderr, "indexlr: unknown option '%s'\n", argv[i]);
            usage(stderr);
            return 2;
        } else if (!in) in = argv[i];
        else {
            fprintf(stderr, "indexlr: more than one input file given\n");
            return 2;
        }
    }
    if (!k || !w || !in) {
        fprintf(stderr, "indexlr: -k, -w and an input FASTA are required\n");
        usage(stderr);
        return 2;
    }
    mxg_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.k = k;
    cfg.w = w;
    cfg.variant = variant;
    cfg.device = device;
    cfg.flags = dense ? MXG_FLAG_DENSE_ONLY : 0;
    cfg.host_threads = threads;  // `-t`: workers that move the FASTA text to the GPU and the TSV text out (reference ntJoin:205)
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    mxg_handle *h = nullptr;
    int rc = mxg_create(&cfg, &h);
    if (rc != MXG_OK) {
        fprintf(stderr, "indexlr: %s\n", mxg_last_error(nullptr));
        return 1;
    }
    const double t1 = now();
    int a = mxg_add_assembly_fasta(h, in, 1.0, in);
    const double t2 = now();
    double t3 = t2;
    if (a < 0 || (rc = mxg_sketch(h, a)) != MXG_OK || (t3 = now(), rc = mxg_write_tsv(h, a, out, with_pos, with_strand, with_seq)) != MXG_OK) {
        fprintf(stderr, "indexlr: %s\n", mxg_last_error(h));
        mxg_destroy(h);
        remove_partial(out);  // leave no partial output behind
        return 1;
    }
    if (verbose) {
        fprintf(stderr, "indexlr: device + handle %.3f s, FASTA -> packed bases in HBM %.3f s, sketch %.3f s, TSV %.3f s\n", t1 - t0, t2 - t1,
                t3 - t2, now() - t3);
        mxg_stats st;
        memset(&st, 0, sizeof st);
        st.struct_size = sizeof st;
        if (mxg_get_stats(h, &st) == MXG_OK)
            fprintf(stderr, "indexlr: %llu bases, %llu valid k-mers, %llu minimizers (%llu candidates, %llu k-mers via dense fix-up)\n",
                    (unsigned long long)st.bases, (unsigned long long)st.kmers, (unsigned long long)st.minimizers,
                    (unsigned long long)st.candidates, (unsigned long long)st.dense_kmers);
    }
    mxg_destroy(h);
    return 0;
}
