"""The scaffold path strings the reference's own tests assert (reference tests/ntjoin_test.py:85,97,104,111,120,133,148,157,165):
`contig ori:start-end gapN ...` as bin/ntjoin_assemble.py:605-607 prints a path.  The coordinates, orientations and gap sizes in
them are what rows f1 (paths) and f4 (format_path inputs) of SURVEY.md 8 must reproduce; they are typed in here from the
reference's test file and pin those rows without the igraph stand-in that generated tests/golden."""
import re

# golden case -> (the -n of the reference's test, the set of path strings it expects)
EXPECTED = {
    "f-f_w1000": (2, {"1_f+:0-1981 20N 2_f+:0-2329"}),                                          # ntjoin_test.py:81-85
    "f-r_w1000": (2, {"1_f+:0-1981 20N 2_r-:0-2329"}),                                          # :93-97
    "r-f_w1000": (2, {"1_r-:0-1981 20N 2_f+:0-2329"}),                                          # :100-104
    "r-r_w1000": (2, {"1_r-:0-1981 20N 2_r-:0-2329"}),                                          # :107-111
    "gap-dist_w500": (1, {"2_1_p+:0-2492 100N 2_2_n-:0-2574", "1_1_p+:0-1744 124N 1_2_p+:0-1844"}),   # :115-122
    "regions-ff-rr_w500": (1, {"2_1n-1_2p-:0-2232 20N 1_1p-2_2n-:2110-4489",
                               "1_1p-2_2n+:0-1568 477N 2_1n-1_2p+:2712-4379"}),                # :128-135
    "regions-fr-rf_w500": (2, {"2_1n-1_2n-:0-2232 253N 1_1p-2_2p+:2058-4489",
                               "1_1p-2_2p+:0-1624 191N 2_1n-1_2n-:2518-4379"}),                # :143-150, 152-159
    "f-f-f_w1000": (1, {"1_f+:0-1981 20N 2_f+:0-2329"}),                                        # :161-165
}


def path_string(nodes):
    """nodes: [contig, ori, start, end, contig_size, first_mx, terminal_mx, gap_size, raw_gap_size] per path node
    (reference bin/path_node.py; untrimmed nodes: adjusted start / end = start / end) -> the line of <prefix>.path"""
    s = " ".join(f"{n[0]}{n[1]}:{n[2]}-{n[3]} {n[7]}N" for n in nodes)
    return re.sub(r"\s+\d+N$", "", s)
