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

// Writes double matrices (uncompressed miMATRIX elements), in the given order.
struct DoubleMatrix {
    std::string name;
    size_t rows, cols;
    const double *data;            // column-major rows x cols
};
bool write(const std::string &path, const std::vector<DoubleMatrix> &vars, std::string &err);

}  // namespace matio
