/* ORACLE (test infrastructure only) -- the reference's UNSTABLE introsort.
 *
 * Restates ksort.h:176-227 (introsort), :155-175 (combsort fallback) and
 * :146-153 (insertion sort).  The permutation produced for equal keys is
 * result-affecting throughout bwa-mem (hard part 1), so
 * the exact sequence of comparisons and swaps is reproduced: median-of-three
 * with the reference's odd middle element, partitions of <=16 elements left to
 * one final insertion sort, combsort when the depth budget is exhausted.
 *
 *   ORA_SORT_DEFINE(name, type, LT)  defines  static void ora_isort_<name>(size_t n, type *a)
 */
#ifndef ORA_SORT_H
#define ORA_SORT_H
#include <stdlib.h>

#define ORA_SORT_DEFINE(name, type_t, LT)                                                     \
static void ora_insertion_##name(type_t *s, type_t *t)                                        \
{                                                                                             \
	type_t *i, *j, tmp;                                                                       \
	for (i = s + 1; i < t; ++i)                                                               \
		for (j = i; j > s && LT(*j, *(j-1)); --j) { tmp = *j; *j = *(j-1); *(j-1) = tmp; }    \
}                                                                                             \
static void ora_comb_##name(size_t n, type_t *a)                                              \
{                                                                                             \
	const double shrink = 1.2473309501039786540366528676643;                                  \
	int swapped;                                                                              \
	size_t gap = n;                                                                           \
	type_t tmp, *i, *j;                                                                       \
	do {                                                                                      \
		if (gap > 2) {                                                                        \
			gap = (size_t)(gap / shrink);                                                     \
			if (gap == 9 || gap == 10) gap = 11;                                              \
		}                                                                                     \
		swapped = 0;                                                                          \
		for (i = a; i < a + n - gap; ++i) {                                                   \
			j = i + gap;                                                                      \
			if (LT(*j, *i)) { tmp = *i; *i = *j; *j = tmp; swapped = 1; }                     \
		}                                                                                     \
	} while (swapped || gap > 2);                                                             \
	if (gap != 1) ora_insertion_##name(a, a + n);                                             \
}                                                                                             \
static void ora_isort_##name(size_t n, type_t *a)                                             \
{                                                                                             \
	struct frame { type_t *lo, *hi; int depth; } *stack, *top;                                \
	int d;                                                                                    \
	type_t pivot, tmp, *s, *t, *i, *j, *k;                                                    \
	if (n < 1) return;                                                                        \
	if (n == 2) { if (LT(a[1], a[0])) { tmp = a[0]; a[0] = a[1]; a[1] = tmp; } return; }      \
	for (d = 2; 1ul << d < n; ++d);                                                           \
	stack = (struct frame*)malloc(sizeof(struct frame) * (sizeof(size_t) * d + 2));           \
	top = stack; s = a; t = a + (n - 1); d <<= 1;                                             \
	for (;;) {                                                                                \
		if (s < t) {                                                                          \
			if (--d == 0) { ora_comb_##name(t - s + 1, s); t = s; continue; }                 \
			i = s; j = t; k = i + ((j - i) >> 1) + 1;                                         \
			if (LT(*k, *i)) { if (LT(*k, *j)) k = j; }                                        \
			else k = LT(*j, *i) ? i : j;                                                      \
			pivot = *k;                                                                       \
			if (k != t) { tmp = *k; *k = *t; *t = tmp; }                                      \
			for (;;) {                                                                        \
				do ++i; while (LT(*i, pivot));                                                \
				do --j; while (i <= j && LT(pivot, *j));                                      \
				if (j <= i) break;                                                            \
				tmp = *i; *i = *j; *j = tmp;                                                  \
			}                                                                                 \
			tmp = *i; *i = *t; *t = tmp;                                                      \
			if (i - s > t - i) {                                                              \
				if (i - s > 16) { top->lo = s; top->hi = i - 1; top->depth = d; ++top; }      \
				s = t - i > 16 ? i + 1 : t;                                                   \
			} else {                                                                          \
				if (t - i > 16) { top->lo = i + 1; top->hi = t; top->depth = d; ++top; }      \
				t = i - s > 16 ? i - 1 : s;                                                   \
			}                                                                                 \
		} else {                                                                              \
			if (top == stack) { free(stack); ora_insertion_##name(a, a + n); return; }        \
			--top; s = top->lo; t = top->hi; d = top->depth;                                  \
		}                                                                                     \
	}                                                                                         \
}

#endif
