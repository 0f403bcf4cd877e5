// Layout of the scratch da_edge_in_sym / da_edge_in_classes fill (csrc/k_edge_io.hip) and da_edge_layer_table reads (csrc/k_edge_layer.hip):
//   [0, 2B)                      flags: [b] symmetric, [B + b] one-hot (1 = the graph took the table path)
//   [in_table_offset(B), ...)    T[(de_in + 2)][de] bf16: the hidden features of the de_in one-hot inputs, of the all-zero input
//                                (row de_in) and of a masked pair (row de_in + 1), in the stored word order of the hidden tensor
//   [in_hist_offset(...), ...)   hist[B][IN_HIST] int: pairs per table row
#pragma once
#include <cstddef>
constexpr int IN_HIST = 16;                          // ints per graph in the class histogram (de_in + 2 <= 10 used)
static inline size_t in_table_offset(int B) { return ((size_t)2 * B + 255) / 256 * 256; }
static inline size_t in_hist_offset(int B, int de_in, int de) { return (in_table_offset(B) + (size_t)(de_in + 2) * de * 2 + 255) / 256 * 256; }
