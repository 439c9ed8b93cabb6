// svdf_stdsort.h -- libstdc++'s std::sort restated for host and device, element for element.
//
// PairwiseRankGenerator::sample_cmp (apex_svd_data.cpp:920-944) sorts a user's rows with std::sort(pos.begin(), pos.end(), cmp_rate)
// (cmp_rate: a.label < b.label, :917-919) and then picks rows BY POSITION in the sorted vector.  std::sort is not stable, rank labels
// are few (0 / 1, 1 ... 5), so which row sits at a position depends on what the library's introsort does with equal keys.  libstdc++
// is a system library, not part of the reference tree; its algorithm (bits/stl_algo.h, unchanged from GCC 4.x to 13: __sort ->
// __introsort_loop with _S_threshold = 16 and depth limit 2 * floor(log2 n), __move_median_to_first on (first + 1, mid, last - 1),
// __unguarded_partition, heap sort via __partial_sort when the depth limit is hit, __final_insertion_sort) is restated here on an array
// of row ids compared through their labels.  The sub-ranges introsort recurses into are disjoint, so the explicit stack below may
// visit them in any order.  tests/test_rank_sampler.py compares this with std::sort itself on the host (svdf_debug_sort_labels);
// the device sampler (svdf_k_sample.hip) runs the same code per user block.
// Provenance: this file follows the ALGORITHM of GNU libstdc++'s <bits/stl_algo.h> (GPLv3 with the GCC Runtime Library Exception 3.1); no text of
// it is copied, the code below is written against the description above.
#ifndef SVDF_STDSORT_H_
#define SVDF_STDSORT_H_

#ifdef __HIPCC__
#define SVDF_HD __host__ __device__ __forceinline__
#else
#define SVDF_HD inline
#endif

namespace svdf {
namespace stdsort {

struct ByLabel {
    const float *label;
    SVDF_HD bool operator()(int a, int b) const { return label[a] < label[b]; }
};

template <typename C> SVDF_HD void iter_swap(int *a, int *b, const C &) { const int t = *a; *a = *b; *b = t; }

template <typename C> SVDF_HD void move_median_to_first(int *result, int *a, int *b, int *c, const C &comp) {
    if (comp(*a, *b)) {
        if (comp(*b, *c)) iter_swap(result, b, comp);
        else if (comp(*a, *c)) iter_swap(result, c, comp);
        else iter_swap(result, a, comp);
    } else if (comp(*a, *c)) iter_swap(result, a, comp);
    else if (comp(*b, *c)) iter_swap(result, c, comp);
    else iter_swap(result, b, comp);
}
template <typename C> SVDF_HD int *unguarded_partition(int *first, int *last, int *pivot, const C &comp) {
    for (;;) {
        while (comp(*first, *pivot)) ++first;
        --last;
        while (comp(*pivot, *last)) --last;
        if (!(first < last)) return first;
        iter_swap(first, last, comp);
        ++first;
    }
}
template <typename C> SVDF_HD void push_heap(int *first, long hole, long top, int value, const C &comp) {
    long parent = (hole - 1) / 2;
    while (hole > top && comp(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
template <typename C> SVDF_HD void adjust_heap(int *first, long hole, long len, int value, const C &comp) {
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (comp(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    push_heap(first, hole, top, value, comp);
}
// __partial_sort(first, last, last): __heap_select (= __make_heap, nothing beyond middle) + __sort_heap
template <typename C> SVDF_HD void heap_sort(int *first, int *last, const C &comp) {
    const long len = last - first;
    if (len >= 2) {
        long parent = (len - 2) / 2;
        for (;;) {
            const int value = first[parent];
            adjust_heap(first, parent, len, value, comp);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) {
        --last;
        const int value = *last;
        *last = *first;
        adjust_heap(first, 0, last - first, value, comp);
    }
}
template <typename C> SVDF_HD void unguarded_linear_insert(int *last, const C &comp) {
    const int val = *last;
    int *next = last;
    --next;
    while (comp(val, *next)) {
        *last = *next;
        last = next;
        --next;
    }
    *last = val;
}
template <typename C> SVDF_HD void insertion_sort(int *first, int *last, const C &comp) {
    if (first == last) return;
    for (int *i = first + 1; i != last; ++i) {
        if (comp(*i, *first)) {
            const int val = *i;
            for (int *p = i; p != first; --p) *p = *(p - 1);   // move_backward(first, i, i + 1)
            *first = val;
        } else {
            unguarded_linear_insert(i, comp);
        }
    }
}
SVDF_HD int floor_log2(long n) {
    int k = 0;
    while (n > 1) { n >>= 1; k++; }
    return k;
}

// std::sort(a, a + n, comp)
template <typename C> SVDF_HD void sort(int *a, long n, const C &comp) {
    if (n <= 0) return;
    // __introsort_loop
    struct Frame { int *first, *last; int depth; };
    Frame stack[72];
    int sp = 0;
    stack[sp++] = Frame{a, a + n, floor_log2(n) * 2};
    while (sp > 0) {
        Frame f = stack[--sp];
        while (f.last - f.first > 16) {
            if (f.depth == 0) { heap_sort(f.first, f.last, comp); f.last = f.first; break; }
            --f.depth;
            int *mid = f.first + (f.last - f.first) / 2;
            move_median_to_first(f.first, f.first + 1, mid, f.last - 1, comp);
            int *cut = unguarded_partition(f.first + 1, f.last, f.first, comp);
            stack[sp++] = Frame{cut, f.last, f.depth};   // __introsort_loop(cut, last, depth_limit)
            f.last = cut;
        }
    }
    // __final_insertion_sort
    if (n > 16) {
        insertion_sort(a, a + 16, comp);
        for (int *i = a + 16; i != a + n; ++i) unguarded_linear_insert(i, comp);
    } else {
        insertion_sort(a, a + n, comp);
    }
}

// std::lower_bound over the sorted ids: first position whose label is not < value
SVDF_HD long lower_bound_label(const int *a, long n, const float *label, float value) {
    long first = 0, len = n;
    while (len > 0) {
        const long half = len >> 1;
        if (label[a[first + half]] < value) { first += half + 1; len -= half + 1; }
        else len = half;
    }
    return first;
}

}  // namespace stdsort
}  // namespace svdf
#endif
