// mxgraph_main.cpp -- FASTA files -> minimizer TSVs + <prefix>.mx.dot in ONE process on the GPU, through the C-ABI of
// libntjoin_mx.so.  It stands for the two recipes of the reference that make up the hot path,
//     indexlr --seq --long --pos -k $(k) -w $(w) -t $(t) $< > $@        (reference ntJoin:204-205, once per assembly)
//     ntjoin_assemble.py ... (minimizer graph part)                      (reference ntJoin:228-230, bin/ntjoin.py:189-204)
// without their seams: one HIP initialisation, the sketches stay in HBM for the graph stage (no TSV is parsed back), and
// the TSVs -- still written, they are ntJoin's checkpoint files -- are formatted on the device while the host
// workers are busy with the .mx.dot text.  Flags follow the reference's ntjoin_assemble.py (-s -l -r -p -k, FILES) plus -w / -t.
// Byte-identical output with `indexlr` per assembly followed by `python -m ntjoin_amd.run` (tests/test_gpu_cli.py).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cerrno>
#include <csignal>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

#include <sys/prctl.h>
#include <sys/stat.h>
#include <sys/wait.h>

#include "ntjoin_mx.h"

// a partial output is removed only when its name is a regular file: a FIFO, /dev/stdout or a process substitution is not ours to unlink
static void remove_partial(const char *path)
{
    struct stat sb;
    if (strcmp(path, "-") != 0 && lstat(path, &sb) == 0 && S_ISREG(sb.st_mode)) remove(path);
}

static void usage(FILE *f)
{
    fputs("Usage: mxgraph -k K -w W -s TARGET.fa [-l TARGET_WEIGHT] -r \"W1 W2 ...\" [-p PREFIX] [-t T] REF1.fa [REF2.fa ...]\n"
          "  sketches every assembly on the GPU, writes <fasta>.k<K>.w<W>.tsv beside each input (indexlr --seq --long --pos format)\n"
          "  and the minimizer graph <PREFIX>.mx.dot (references first, in the order given, then the target)\n"
          "  -k K            k-mer size (required)\n"
          "  -w W            window size in k-mers (required)\n"
          "  -s FASTA        the assembly to be scaffolded (required)\n"
          "  -l X            its edge weight (default 1)\n"
          "  -r \"X ...\"      one edge weight per reference (required)\n"
          "  -p PREFIX       output prefix (default out)\n"
          "  -t T            host worker threads (FASTA in, TSV / dot text out; default 4)\n"
          "  --variant v2|v1 canonical hash: v2 = fwd+rev (current btllib, default), v1 = min(fwd,rev)\n"
          "  --device N      HIP device ordinal\n"
          "  --no-tsv        do not write the TSV checkpoints\n"
          "  -v              timings and statistics on stderr\n",
          f);
}

static bool opt_val(int argc, char **argv, int &i, const char *name, const char **val)
{
    const size_t n = strlen(name);
    if (strncmp(argv[i], name, n) != 0) return false;
    if (argv[i][n] == 0) {
        if (i + 1 >= argc) return false;
        *val = argv[++i];
        return true;
    }
    if (name[1] != '-') {
        *val = argv[i] + n;
        return true;
    }
    if (argv[i][n] == '=') {
        *val = argv[i] + n + 1;
        return true;
    }
    return false;
}

// The work is done by a child of the process the user started; the parent leaves with the child's status the moment the child says
// that every output is written and closed.  What the child still has to do then -- hand 13 GB of HBM, its pinned buffers and its
// queues back to the driver -- took 0.11-0.15 s between the end of main() and the caller's wait() in most runs at 3 Gbp + 3 Gbp
// (0.001 s in others, whatever was freed beforehand): nobody needs to wait for that.  MXG_NO_DETACH=1: one process (sanitizer
// runs, whose reports come with the real exit; debuggers).
static int g_done_fd = -1;
static void report_done(int status)
{
    fflush(stdout);
    fflush(stderr);
    if (g_done_fd >= 0) {
        const unsigned char b = (unsigned char)status;
        // (a caller that reads this process's output through pipes waits for their last writer: the worker lets go of them too)
        const int nul = open("/dev/null", O_RDWR);
        if (nul >= 0) {
            dup2(nul, 0);
            dup2(nul, 1);
            dup2(nul, 2);
            if (nul > 2) close(nul);
        }
        if (write(g_done_fd, &b, 1) != 1) { /* the parent is gone: nothing to tell */ }
        close(g_done_fd);
        g_done_fd = -1;
    }
}

static int run(int argc, char **argv)
{
    unsigned k = 0, w = 0, threads = 4;
    unsigned variant = MXG_VARIANT_V2_SUM;
    int device = -1, verbose = 0, no_tsv = 0;
    double target_weight = 1.0;
    const char *target = nullptr, *weights_arg = nullptr, *v = nullptr;
    std::string prefix = "out";
    std::vector<std::string> refs;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--help") || !strcmp(argv[i], "-h")) { usage(stdout); return 0; }
        else if (!strcmp(argv[i], "-v")) verbose = 1;
        else if (!strcmp(argv[i], "--no-tsv")) no_tsv = 1;
        else if (opt_val(argc, argv, i, "--variant", &v)) variant = (!strcmp(v, "v1") || !strcmp(v, "min")) ? MXG_VARIANT_V1_MIN : MXG_VARIANT_V2_SUM;
        else if (opt_val(argc, argv, i, "--device", &v)) device = atoi(v);
        else if (opt_val(argc, argv, i, "-k", &v)) k = (unsigned)strtoul(v, nullptr, 10);
        else if (opt_val(argc, argv, i, "-w", &v)) w = (unsigned)strtoul(v, nullptr, 10);
        else if (opt_val(argc, argv, i, "-t", &v)) threads = (unsigned)strtoul(v, nullptr, 10);
        else if (opt_val(argc, argv, i, "-s", &v)) target = v;
        else if (opt_val(argc, argv, i, "-l", &v)) target_weight = atof(v);
        else if (opt_val(argc, argv, i, "-r", &v)) weights_arg = v;
        else if (opt_val(argc, argv, i, "-p", &v)) prefix = v;
        else if (argv[i][0] == '-' && argv[i][1] != 0) {
            fprintf(stderr, "mxgraph: unknown option '%s'\n", argv[i]);
            usage(stderr);
            return 2;
        } else refs.push_back(argv[i]);
    }
    if (!k || !w || !target || !weights_arg || refs.empty()) {
        fprintf(stderr, "mxgraph: -k, -w, -s, -r and at least one reference FASTA are required\n");
        usage(stderr);
        return 2;
    }
    std::vector<double> weights;
    {
        char *end = nullptr;
        for (const char *p = weights_arg; *p;) {
            while (*p == ' ' || *p == '\t') ++p;
            if (!*p) break;
            const double x = strtod(p, &end);
            if (end == p) {
                fprintf(stderr, "mxgraph: cannot read the weights '%s'\n", weights_arg);
                return 2;
            }
            weights.push_back(x);
            p = end;
        }
    }
    if (weights.size() != refs.size()) {  // (the reference's check, bin/ntjoin_assemble.py:788-797: exit status 1)
        printf("ERROR: -r lists %zu weight(s) but %zu reference file(s) were given; there must be exactly one weight per reference, "
               "in the same order.\n", weights.size(), refs.size());
        return 1;
    }
    mxg_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.k = k;
    cfg.w = w;
    cfg.variant = variant;
    cfg.device = device;
    cfg.host_threads = threads;
    if (!getenv("MXG_KEEP_BUFFERS")) cfg.flags |= MXG_FLAG_ONE_SHOT;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto wall = []() { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); };
    if (getenv("MXG_DEBUG_IO")) fprintf(stderr, "[mxg] main() entered at %.3f (epoch seconds)\n", wall());
    const double t0 = now();
    mxg_handle *h = nullptr;
    if (mxg_create(&cfg, &h) != MXG_OK) {
        fprintf(stderr, "mxgraph: %s\n", mxg_last_error(nullptr));
        return 1;
    }
    const double t1 = now();
    // references in the order given, the target last: the reference's load order (bin/ntjoin.py:178-186)
    std::vector<std::string> fastas = refs, tsvs;
    fastas.push_back(target);
    weights.push_back(target_weight);
    const std::string suffix = ".k" + std::to_string(k) + ".w" + std::to_string(w) + ".tsv";
    auto fail = [&](const char *what) {
        fprintf(stderr, "mxgraph: %s: %s\n", what, mxg_last_error(h));
        mxg_destroy(h);
        return 1;
    };
    for (size_t i = 0; i < fastas.size(); ++i) {
        tsvs.push_back(fastas[i] + suffix);  // (the assembly's name in the graph is its TSV's, as on the two-process route)
        if (mxg_add_assembly_fasta(h, tsvs[i].c_str(), weights[i], fastas[i].c_str()) < 0) return fail(fastas[i].c_str());
    }
    const double t2 = now();
    if (mxg_sketch(h, MXG_SKETCH_ALL) != MXG_OK) return fail("sketch");
    const double t3 = now();
    if (mxg_build_graph(h) != MXG_OK) return fail("graph");
    const double t4 = now();
    const std::string dot = prefix + ".mx.dot";
    // the .mx.dot text (host workers) and the TSVs (formatted on the device) at the same time
    std::vector<const char *> tsv_ptrs;
    for (auto &t : tsvs) tsv_ptrs.push_back(no_tsv ? nullptr : t.c_str());
    if (mxg_write_outputs(h, dot.c_str(), tsv_ptrs.data(), 1, 0, 1) != MXG_OK) {
        remove_partial(dot.c_str());
        if (!no_tsv)
            for (auto &t : tsvs) remove_partial(t.c_str());  // leave no partial output behind
        return fail("writing the outputs");
    }
    const double t5 = now();
    if (verbose) {
        fprintf(stderr, "mxgraph: device + handle %.3f s, FASTA -> packed bases in HBM %.3f s, sketches %.3f s, graph %.3f s, %s + TSVs %.3f s\n",
                t1 - t0, t2 - t1, t3 - t2, t4 - t3, dot.c_str(), t5 - t4);
        mxg_stats st;
        memset(&st, 0, sizeof st);
        st.struct_size = sizeof st;
        const double t6 = now();
        if (mxg_get_stats(h, &st) == MXG_OK)
            fprintf(stderr, "mxgraph: %llu bases, %llu minimizers, %llu vertices, %llu edges\n", (unsigned long long)st.bases,
                    (unsigned long long)st.minimizers, (unsigned long long)st.vertices, (unsigned long long)st.edges);
        if (getenv("MXG_DEBUG_IO")) fprintf(stderr, "[mxg] statistics for -v: %.3f s\n", now() - t6);
    }
    // every output is complete and closed; the process ends without returning tens of GB of HBM buffer by buffer first (the
    // driver reclaims them with the process: 0.1-0.2 s of a 1.3 s run at 3 Gbp + 3 Gbp)
    if (getenv("MXG_DEBUG_IO")) fprintf(stderr, "[mxg] leaving main() at %.3f (epoch seconds)\n", wall());
    report_done(0);
    if (!getenv("MXG_CLEAN_EXIT")) _exit(0);
    mxg_destroy(h);
    return 0;
}

int main(int argc, char **argv)
{
    int fds[2];
    if (getenv("MXG_NO_DETACH") || pipe(fds) != 0) return run(argc, argv);
    fflush(stdout);
    fflush(stderr);
    const pid_t parent_pid = getpid();
    const pid_t pid = fork();  // (before anything touches the GPU: a HIP context does not survive a fork)
    if (pid < 0) {
        close(fds[0]);
        close(fds[1]);
        return run(argc, argv);
    }
    if (pid == 0) {  // the worker
        // (it lives and dies with the process the user started: a parent that is killed takes it along -- after the parent has left
        // in the regular way, the signal finds the worker with nothing left to do but hand its memory back)
        prctl(PR_SET_PDEATHSIG, SIGTERM);
        if (getppid() != parent_pid) _exit(143);
        close(fds[0]);
        g_done_fd = fds[1];
        if (const char *sig = getenv("MXG_TEST_WORKER_SIGNAL")) raise(atoi(sig));  // (test knob: a worker that dies without a word)
        const int rc = run(argc, argv);  // (the good end reports from inside and leaves through _exit)
        report_done(rc);
        _exit(rc);
    }
    close(fds[1]);
    unsigned char b = 0;
    ssize_t got;
    do got = read(fds[0], &b, 1);
    while (got < 0 && errno == EINTR);
    if (got == 1) return (int)b;  // every output is complete; the worker finishes on its own
    int st = 0;                   // the worker ended without a word: its status is the answer
    while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {
    }
    if (WIFEXITED(st)) return WEXITSTATUS(st);
    return 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
}
