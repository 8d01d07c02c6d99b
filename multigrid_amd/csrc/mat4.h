// mat4.h -- MATLAB Level-4 (".mat -v4") reader/writer for the command-line
// driver, covering what the reference's Input.mat / Output.mat contract uses.
//
// Record layout (reference DTSource/DTMatlabDataFile.cpp:28-35,219-283,456-492):
//   int32 type, mrows, ncols, imagf, namelen; char name[namelen] (NUL included);
//   mrows*ncols elements, column-major.
//   type = M*1000 + O*100 + P*10 + T:  M byte order (0 little, 1 big),
//   P element type (0 double, 1 float, 2 int32, 3 int16, 4 uint16, 5 uint8),
//   T 0 full matrix / 1 text.
// Like the reference's reader, the byte order is decided on the first record
// (a type outside 0..52 means the file has the other order) and records of
// unknown element type end the scan.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace mat4 {

struct Entry {
    int m = 0, n = 0, type = 0;
    long location = -1;  // file offset of the data
};

inline void swap4(unsigned char *p, size_t nwords)
{
    for (size_t k = 0; k < nwords; k++, p += 4) {
        unsigned char t = p[0]; p[0] = p[3]; p[3] = t;
        t = p[1]; p[1] = p[2]; p[2] = t;
    }
}
inline void swapN(unsigned char *p, size_t count, int width)
{
    for (size_t k = 0; k < count; k++, p += width)
        for (int a = 0, b = width - 1; a < b; a++, b--) { unsigned char t = p[a]; p[a] = p[b]; p[b] = t; }
}

class Reader {
public:
    explicit Reader(const std::string &path) : path_(path)
    {
        fp_ = fopen(path.c_str(), "rb");
        if (!fp_) return;
        static const int sizes[10] = {8, 4, 4, 2, 2, 1, -1, -1, -1, -1};
        long file_size = 0;
        if (fseek(fp_, 0, SEEK_END) == 0) file_size = ftell(fp_);
        if (fseek(fp_, 0, SEEK_SET) != 0 || file_size < 0) {
            error_ = "cannot seek in the file";
            return;
        }
        for (;;) {
            const long starts = ftell(fp_);
            int32_t h[5];
            if (fread(h, sizeof(h), 1, fp_) != 1) break;  // end of file
            if (starts == 0 && (h[0] < 0 || h[0] > 52)) {
                swap4((unsigned char *)h, 5);
                swap_ = true;
            } else if (swap_) {
                swap4((unsigned char *)h, 5);
            }
            if (h[4] <= 0 || h[4] > 255 || h[0] < 0 || h[0] > 1052) {
                error_ = "Invalid file format, make sure you used the -v4 in matlab.";
                break;
            }
            const int type = h[0] % 1000;
            char name[256];
            if (fread(name, 1, (size_t)h[4], fp_) != (size_t)h[4]) break;
            name[h[4] - 1] = 0;
            if (sizes[(type % 100) / 10] < 0) {
                error_ = "Unexpected variable type in the file.";
                break;
            }
            // dimensions come from the file: negative ones, or more data than the file holds,
            // end the scan (the sizes below would otherwise wrap or ask for absurd amounts)
            const long remaining = file_size - (starts + 20 + h[4]);
            const long width = sizes[(type % 100) / 10];
            if (h[1] < 0 || h[2] < 0 || remaining < 0 ||
                (h[1] > 0 && h[2] > 0 && ((long)h[1] > remaining / width || (long)h[2] > remaining / width / (long)h[1]))) {
                error_ = "A variable's dimensions do not fit the file.";
                break;
            }
            const long esize = width * h[1] * h[2];
            Entry e;
            e.m = h[1];
            e.n = h[2];
            e.type = type;
            e.location = starts + 20 + h[4];
            entries_[name] = e;
            if (fseek(fp_, starts + esize + 20 + h[4], SEEK_SET) != 0) break;
        }
    }
    ~Reader() { if (fp_) fclose(fp_); }
    bool is_open() const { return fp_ != nullptr; }
    const std::string &error() const { return error_; }
    bool contains(const std::string &name) const { return entries_.count(name) != 0; }

    // Reads a numeric variable as doubles (column-major).  Returns false if absent.
    bool read_double(const std::string &name, std::vector<double> &out, int &m, int &n)
    {
        auto it = entries_.find(name);
        if (it == entries_.end()) return false;
        const Entry &e = it->second;
        m = e.m;
        n = e.n;
        const size_t cnt = (size_t)e.m * (size_t)e.n;
        out.assign(cnt, 0.0);
        if (cnt == 0) return true;
        static const int sizes[6] = {8, 4, 4, 2, 2, 1};
        const int P = (e.type % 100) / 10;
        const int w = sizes[P];
        std::vector<unsigned char> raw(cnt * (size_t)w);
        if (fseek(fp_, e.location, SEEK_SET) != 0 || fread(raw.data(), (size_t)w, cnt, fp_) != cnt) return false;
        if (swap_ && w > 1) swapN(raw.data(), cnt, w);
        for (size_t k = 0; k < cnt; k++) {
            const unsigned char *p = raw.data() + k * (size_t)w;
            switch (P) {
            case 0: { double v; memcpy(&v, p, 8); out[k] = v; break; }
            case 1: { float v; memcpy(&v, p, 4); out[k] = v; break; }
            case 2: { int32_t v; memcpy(&v, p, 4); out[k] = v; break; }
            case 3: { int16_t v; memcpy(&v, p, 2); out[k] = v; break; }
            case 4: { uint16_t v; memcpy(&v, p, 2); out[k] = v; break; }
            default: out[k] = *p; break;
            }
        }
        return true;
    }

private:
    std::string path_, error_;
    FILE *fp_ = nullptr;
    bool swap_ = false;
    std::map<std::string, Entry> entries_;
};

// Appends one double matrix (little-endian, type 0) the way
// DTMatlabDataFile::Save(const DTDoubleArray&, name) does (:456-492).
class Writer {
public:
    explicit Writer(const std::string &path) { fp_ = fopen(path.c_str(), "wb"); }
    ~Writer() { if (fp_) fclose(fp_); }
    bool is_open() const { return fp_ != nullptr; }
    bool save(const double *a, int m, int n, const std::string &name)
    {
        if (!fp_) return false;
        int32_t h[5] = {0, m, n, 0, (int32_t)(name.size() + 1)};
        if (fwrite(h, sizeof(h), 1, fp_) != 1) return false;
        if (fwrite(name.c_str(), 1, name.size() + 1, fp_) != name.size() + 1) return false;
        const size_t cnt = (size_t)m * (size_t)n;
        return cnt == 0 || fwrite(a, sizeof(double), cnt, fp_) == cnt;
    }

private:
    FILE *fp_ = nullptr;
};

} // namespace mat4
