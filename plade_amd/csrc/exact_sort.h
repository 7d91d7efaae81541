// plade_amd/csrc/exact_sort.h -- the host's one large sort: the clusters of candidate transforms by size, descending
// (code/PLADE/util.cpp:335-345: std::sort(sortVec.begin(), sortVec.end(), myCompareGreater), util.h:347-365).
//
// std::sort is not stable and a registration has ~30 000 clusters of which most have size 1 or 2, so WHICH of the equally
// large clusters comes first -- and with it the order of the candidate transforms -- is decided by the algorithm itself:
// libstdc++'s introsort (bits/stl_algo.h: median of three to the front, unguarded Hoare partition, recursion on the right
// part, heap sort below depth 2 lg n, one final insertion sort over ranges of <= 16).  A drop-in has to produce that very
// permutation.  Calling std::sort does (the oracle does), at ~0.5 ms of host time per registration: on tied keys every
// comparison of the partition loop is a coin flip for the branch predictor.  This is the same algorithm, step for step
// and swap for swap, with the two scans of the partition collecting their stopping cells block-wise without branches
// (the idea of BlockQuicksort, Edelkamp & Weiss 2016) and exchanging them pair by pair in the sequential order:
//
//   sequential: f stops at the next cell with !(x > p), l at the next cell (downwards) with !(p > x); if !(f < l) return f;
//               swap, ++f.
//   Let L_1 < L_2 < ... be the cells with !(x > p) and R_1 > R_2 > ... those with !(p > x) in the range AS IT WAS at
//   entry.  Swap k-1 leaves a value <= p in cell R_{k-1} and a value >= p in cell L_{k-1} and touches nothing in between,
//   so in round k the scans stop at f_k = min(L_k, R_{k-1}) and l_k = max(R_k, L_{k-1}); f_k < l_k only if f_k = L_k and
//   l_k = R_k.  The rounds therefore exchange L_k with R_k while L_k < R_k and return the first f_k that is not below l_k.
//
// tests/test_host_logic.py compares the permutation with std::sort's (plade_diag_cluster_order) on random, tied, constant,
// sorted and reversed inputs, and both partitions with each other at a reduced depth limit (the heap-sort branch).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>

namespace plade {
namespace exact_sort {

struct Item { float length; int index; };   // util.h:347-350
inline bool greater(const Item &a, const Item &b) { return a.length > b.length; }   // util.h:360-365

constexpr long THRESHOLD = 16;   // _S_threshold
constexpr int BLK = 128;

inline void move_median_to_first(Item *r, Item *a, Item *b, Item *c) {
    if (greater(*a, *b)) {
        if (greater(*b, *c)) std::swap(*r, *b);
        else if (greater(*a, *c)) std::swap(*r, *c);
        else std::swap(*r, *a);
    } else if (greater(*a, *c)) std::swap(*r, *a);
    else if (greater(*b, *c)) std::swap(*r, *c);
    else std::swap(*r, *b);
}

// __unguarded_partition(first, last, pivot) as written in the library
inline Item *partition_sequential(Item *first, Item *last, const Item *pivot) {
    const float p = pivot->length;
    for (;;) {
        while (first->length > p) ++first;
        --last;
        while (p > last->length) --last;
        if (!(first < last)) return first;
        std::swap(*first, *last);
        ++first;
    }
}

// the same exchanges and the same return value, stopping cells collected per block of BLK cells
inline Item *partition_blocked(Item *first, Item *last, const Item *pivot) {
    const float p = pivot->length;
    uint8_t off_l[BLK], off_r[BLK];
    int n_l = 0, n_r = 0, s_l = 0, s_r = 0;      // stopping cells buffered / consumed
    Item *l_scan = first, *r_scan = last;        // classified so far: [first, l_scan) from the left, [r_scan, last) from the right
    Item *l_base = first, *r_base = last;
    Item *l_prev = nullptr, *r_prev = nullptr;   // the cells of the last exchange
    for (;;) {
        // cells from r_prev on (below l_prev + 1) no longer hold what they held at entry: the scans classify up to there only
        Item *lim_l = r_prev ? r_prev : last;
        while (s_l == n_l && l_scan < lim_l) {
            long cnt = lim_l - l_scan;
            if (cnt > BLK) cnt = BLK;
            l_base = l_scan; n_l = 0; s_l = 0;
            for (long i = 0; i < cnt; ++i) { off_l[n_l] = (uint8_t)i; n_l += !(l_base[i].length > p); }
            l_scan += cnt;
        }
        Item *lim_r = l_prev ? l_prev + 1 : first;
        while (s_r == n_r && r_scan > lim_r) {
            long cnt = r_scan - lim_r;
            if (cnt > BLK) cnt = BLK;
            r_base = r_scan; n_r = 0; s_r = 0;
            for (long i = 0; i < cnt; ++i) { off_r[n_r] = (uint8_t)i; n_r += !(p > r_base[-1 - i].length); }
            r_scan -= cnt;
        }
        // all pairs both buffers hold at once, if the last of them is still a proper pair (the left cells ascend, the right
        // cells descend: then every one of them is)
        const int m = std::min(n_l - s_l, n_r - s_r);
        if (m > 0 && l_base + off_l[s_l + m - 1] < r_base - 1 - off_r[s_r + m - 1]) {
            for (int q = 0; q < m; ++q) std::swap(l_base[off_l[s_l + q]], r_base[-1 - (long)off_r[s_r + q]]);
            s_l += m; s_r += m;
            l_prev = l_base + off_l[s_l - 1];
            r_prev = r_base - 1 - off_r[s_r - 1];
            continue;
        }
        Item *f = s_l < n_l ? l_base + off_l[s_l] : lim_l;          // min(L_k, R_{k-1})
        if (f > lim_l) f = lim_l;
        Item *l = s_r < n_r ? r_base - 1 - off_r[s_r] : lim_r - 1;  // max(R_k, L_{k-1}); before any exchange the pivot in front
        if (l < lim_r - 1) l = lim_r - 1;                           // of `first` stops the scan
        if (!(f < l)) return f;
        std::swap(*f, *l);
        l_prev = f; r_prev = l; ++s_l; ++s_r;
    }
}

inline void insertion_sort(Item *first, Item *last) {   // __insertion_sort
    if (first == last) return;
    for (Item *i = first + 1; i != last; ++i) {
        const Item val = *i;
        if (greater(val, *first)) { std::memmove(first + 1, first, (size_t)((char *)i - (char *)first)); *first = val; }
        else { Item *j = i; while (greater(val, *(j - 1))) { *j = *(j - 1); --j; } *j = val; }
    }
}
inline void unguarded_insertion_sort(Item *first, Item *last) {   // __unguarded_insertion_sort
    for (Item *i = first; i != last; ++i) {
        const Item val = *i;
        Item *j = i;
        while (greater(val, *(j - 1))) { *j = *(j - 1); --j; }
        *j = val;
    }
}

template <bool BLOCKED>
inline void introsort_loop(Item *first, Item *last, long depth_limit) {
    while (last - first > THRESHOLD) {
        if (depth_limit == 0) { std::partial_sort(first, last, last, greater); return; }   // __partial_sort(first, last, last)
        --depth_limit;
        Item *mid = first + (last - first) / 2;
        move_median_to_first(first, first + 1, mid, last - 1);
        Item *cut = (BLOCKED && last - first >= 3 * BLK) ? partition_blocked(first + 1, last, first)
                                                         : partition_sequential(first + 1, last, first);
        introsort_loop<BLOCKED>(cut, last, depth_limit);
        last = cut;
    }
}

// std::sort(first, last, myCompareGreater); depth_limit < 0: the library's 2 * floor(lg n)
template <bool BLOCKED = true>
inline void sort_descending(Item *first, Item *last, long depth_limit = -1) {
    if (first == last) return;
    const long n = last - first;
    if (depth_limit < 0) {
        long lg = 0;
        while ((1L << (lg + 1)) <= n) ++lg;
        depth_limit = 2 * lg;
    }
    introsort_loop<BLOCKED>(first, last, depth_limit);
    if (n > THRESHOLD) { insertion_sort(first, first + THRESHOLD); unguarded_insertion_sort(first + THRESHOLD, last); }
    else insertion_sort(first, last);
}

}  // namespace exact_sort
}  // namespace plade
