// Minimal MAT-file (Level 5) reader/writer: real numeric arrays only, which is all
// the reference exchanges through MATLAB's libmat (matGetVariable of "DataSet" and
// "labels", gcn_sgd_fp64.cu:1025-1038; matPutVariable of seven double matrices,
// :901-931).  Handles miCOMPRESSED elements (zlib), every numeric storage type and
// the small-element tag format; rejects v7.3 (HDF5) files with a clear message.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace matio {

struct Array {
    std::string name;
    std::vector<size_t> dims;      // column-major, MATLAB order
    bool is_single = false;        // mxSINGLE_CLASS (else converted to double)
    std::vector<double> f64;       // values when !is_single
    std::vector<float> f32;        // values when is_single
    size_t numel() const
    {
        size_t n = 1;
        for (size_t d : dims) n *= d;
        return dims.empty() ? 0 : n;
    }
};

// Reads the named variables (all variables if `names` is empty).  Returns false and
// sets err on failure (missing file, unsupported format, truncated data).  A name
// that is not in the file is simply absent from `out`.
bool read(const std::string &path, const std::vector<std::string> &names, std::vector<Array> &out,
          std::string &err);

// One numeric variable read front to back WITHOUT holding the file or the inflated element in
// memory (the 610 x 340 x 103 Pavia-U DataSet is 171 MB as double, a 1024 x 1024 x 224 scene as
// single 940 MB): open() walks the top-level elements (a compressed element is only inflated as far
// as its name), stops at the variable's data and next() hands out the following `count` values,
// converted to T, through a bounded buffer.  Limit of the format, not of this reader: a MAT v5 data
// element carries a 32-bit byte count, so one variable holds less than 4 GiB (v7.3 / HDF5 files are
// rejected); a larger scene has to be split into several variables or files by band range.
class Stream {
public:
    Stream();
    ~Stream();
    Stream(const Stream &) = delete;
    Stream &operator=(const Stream &) = delete;
    // false + err when the file cannot be read, is not MAT v5, or has no numeric real variable `name`
    bool open(const std::string &path, const std::string &name, std::string &err);
    const std::vector<size_t> &dims() const { return dims_; }
    size_t numel() const { return numel_; }
    bool is_single() const { return is_single_; }
    size_t remaining() const { return numel_ - done_; }
    // the next `count` values in storage (column-major) order; false + err past the end or on bad data
    bool next(float *dst, size_t count, std::string &err);
    bool next(double *dst, size_t count, std::string &err);
    void close();

private:
    struct Impl;
    Impl *im_;
    std::vector<size_t> dims_;
    size_t numel_ = 0, done_ = 0;
    bool is_single_ = false;
    template <typename D>
    bool next_impl(D *dst, size_t count, std::string &err);
};

// Writes double matrices (uncompressed miMATRIX elements), in the given order.
struct DoubleMatrix {
    std::string name;
    size_t rows, cols;
    const double *data;            // column-major rows x cols
};
bool write(const std::string &path, const std::vector<DoubleMatrix> &vars, std::string &err);

}  // namespace matio
