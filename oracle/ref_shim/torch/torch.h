// TEST INFRASTRUCTURE.  Minimal stand-in for the one libtorch type the reference's kernel headers use:
// torch::PackedTensorAccessor32<T, N, RestrictPtrTraits> (stride-aware N-d view with operator[] chaining).
#pragma once
#include <cstddef>
#include <initializer_list>
#include <type_traits>
#include <cstdint>

namespace torch {
template <typename T>
struct RestrictPtrTraits {
  typedef T* PtrType;
};

template <typename T, size_t N>
struct SubAccessor {
  T* data;
  const int* sizes;
  const int* strides;
  SubAccessor<T, N - 1> operator[](int i) const { return SubAccessor<T, N - 1>{data + (int64_t)i * strides[0], sizes + 1, strides + 1}; }
  int size(int d) const { return sizes[d]; }
};
template <typename T>
struct SubAccessor<T, 1> {
  T* data;
  const int* sizes;
  const int* strides;
  T& operator[](int i) const { return data[(int64_t)i * strides[0]]; }
  int size(int d) const { return sizes[d]; }
};

template <typename T, size_t N, template <typename U> class Traits = RestrictPtrTraits, typename index_t = int>
struct PackedTensorAccessor32 {
  T* data;
  int sizes_[N];
  int strides_[N];
  PackedTensorAccessor32() : data(nullptr) {}
  // contiguous view
  PackedTensorAccessor32(T* d, std::initializer_list<int> shape) : data(d) {
    size_t k = 0;
    for (int s : shape) sizes_[k++] = s;
    int st = 1;
    for (int i = (int)N - 1; i >= 0; i--) {
      strides_[i] = st;
      st *= sizes_[i];
    }
  }
  int size(int d) const { return sizes_[d]; }
  template <size_t M = N>
  typename std::enable_if<(M > 1), SubAccessor<T, N - 1>>::type operator[](int i) const {
    return SubAccessor<T, N - 1>{data + (int64_t)i * strides_[0], sizes_ + 1, strides_ + 1};
  }
  template <size_t M = N>
  typename std::enable_if<(M == 1), T&>::type operator[](int i) const {
    return data[(int64_t)i * strides_[0]];
  }
};
}  // namespace torch
