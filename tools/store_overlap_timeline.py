"""Do the Store kernels of the worker thread run BESIDE the Regrid kernels of the caller's stream?  Reads a rocprofv3 --kernel-trace
CSV (*_kernel_trace.csv: start / end timestamps per dispatch) of `bench.py --leg job` and reports, for every cold pass (a cluster of
Store kernels), the time covered by Store kernels, by Regrid kernels within that window, and by both at once.
    python tools/store_overlap_timeline.py <kernel_trace.csv>"""
import csv
import sys

STORE = ("k_tri_", "k_nb_", "k_conserve", "k_grid_bilinear", "k_points_ij", "k_csr_", "k_scan", "k_mark", "k_compact", "k_pyr", "k_bvh", "k_sort", "k_cell_areas",
         "k_max_valence", "k_vertex_range", "rocprim", "k_sum", "k_fan", "k_dual")
APPLY = ("k_apply", "k_wind", "k_rotate", "k_pole", "k_lfu_build")


def base(name):
    return name[5:] if name.startswith("void ") else name


def merge(iv):
    out = []
    for a, b in sorted(iv):
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def total(iv):
    return sum(b - a for a, b in iv)


def inter(x, y):
    i = j = tot = 0
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if b > a:
            tot += b - a
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ks = sorted(((base(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows), key=lambda t: t[1])
    st = [(a, b) for n, a, b, _ in ks if n.startswith(STORE)]
    clusters, cur = [], []
    for a, b in st:
        if cur and a - cur[-1][1] > float(sys.argv[3] if len(sys.argv) > 3 else 20e6):      # cold passes are tens of ms apart (warm passes and set-up lie between)
            clusters.append(cur)
            cur = []
        cur.append((a, b))
    if cur:
        clusters.append(cur)
    for c in clusters:
        lo, hi = c[0][0], c[-1][1]
        ap = [(max(a, lo), min(b, hi)) for n, a, b, _ in ks if n.startswith(APPLY) and b > lo and a < hi]
        ms, ma = merge(c), merge(ap)
        queues = sorted({q for n, a, b, q in ks if a >= lo and b <= hi})
        if len(sys.argv) > 2:      # --dump: every kernel of the window, one line each (ms from the window's start, duration, queue)
            for n, a, b, q in ks:
                if b > lo and a < hi + 3e6:
                    print("%9.3f %8.3f q%s %s" % ((a - lo) / 1e6, (b - a) / 1e6, q, n[:70]))
        print("window %8.3f ms | %3d Store kernels busy %7.3f ms | Regrid kernels busy %7.3f ms | both at once %7.3f ms | queues %s" % (
            (hi - lo) / 1e6, len(c), total(ms) / 1e6, total(ma) / 1e6, inter(ms, ma) / 1e6, ",".join(queues)))


if __name__ == "__main__":
    main()
