// generateDataset -- outlier injector with the command line and the output of the reference's
// scripts/generateDataset.py, so benchmark inputs can be produced without
// Python.  For a given --seed the output is byte-identical to the script's: the draws go through
// a restatement of CPython's `random` module (MT19937 seeded by init_by_array, randint by
// rejection on getrandbits, gauss with the cached second deviate), numbers are printed with
// Python's repr rules, and the script's quirks are kept (information copied from the first
// non-odometry edge when --information is absent, :176-182; 3-D quaternion written as w x y z,
// :101,225,239; v2 bumped to v1+2 when adjacent, :205-206).
//
//   generateDataset -i in.g2o -o out.g2o -n 1000 --seed 7 [-g 2] [-l] [-p] [--information=42,0,0,42,0,42]
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

namespace {

// CPython calls libm's sin() and cos() separately; when both are taken of one argument the
// optimiser would emit a fused sincos(), which does not always round the same way.
__attribute__((noinline)) double sin_sep(double x) { return std::sin(x); }
__attribute__((noinline)) double cos_sep(double x) { return std::cos(x); }

// ---- CPython's Mersenne Twister front end (Modules/_randommodule.c, Lib/random.py) ----------
class PyRandom {
public:
    void seed(uint64_t a)
    {
        // random_seed(): the absolute value as little-endian 32-bit words, at least one
        std::vector<uint32_t> key;
        do { key.push_back((uint32_t)(a & 0xffffffffu)); a >>= 32; } while (a);
        init_by_array(key);
        has_gauss_ = false;
    }
    void seed_from_entropy()
    {
        std::random_device rd;
        std::vector<uint32_t> key(4);
        for (auto& k : key) k = rd();
        init_by_array(key);
        has_gauss_ = false;
    }
    double random()                                   // genrand_res53
    {
        const uint32_t a = next_u32() >> 5, b = next_u32() >> 6;
        return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
    }
    long randint(long lo, long hi) { return lo + (long)randbelow((uint64_t)(hi - lo + 1)); }
    double gauss(double mu, double sigma)
    {
        double z;
        if (has_gauss_) { z = gauss_next_; has_gauss_ = false; }
        else {
            const double x2pi = random() * (2.0 * M_PI);
            const double g2rad = std::sqrt(-2.0 * std::log(1.0 - random()));
            z = cos_sep(x2pi) * g2rad;
            gauss_next_ = sin_sep(x2pi) * g2rad;
            has_gauss_ = true;
        }
        return mu + z * sigma;
    }

private:
    static constexpr int N = 624, M = 397;
    uint32_t mt_[N];
    int idx_ = N + 1;
    bool has_gauss_ = false;
    double gauss_next_ = 0.0;

    void init_genrand(uint32_t s)
    {
        mt_[0] = s;
        for (int i = 1; i < N; ++i) mt_[i] = 1812433253u * (mt_[i - 1] ^ (mt_[i - 1] >> 30)) + (uint32_t)i;
        idx_ = N;
    }
    void init_by_array(const std::vector<uint32_t>& key)
    {
        init_genrand(19650218u);
        int i = 1, j = 0;
        const int klen = (int)key.size();
        for (int k = N > klen ? N : klen; k; --k) {
            mt_[i] = (mt_[i] ^ ((mt_[i - 1] ^ (mt_[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            ++i; ++j;
            if (i >= N) { mt_[0] = mt_[N - 1]; i = 1; }
            if (j >= klen) j = 0;
        }
        for (int k = N - 1; k; --k) {
            mt_[i] = (mt_[i] ^ ((mt_[i - 1] ^ (mt_[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            ++i;
            if (i >= N) { mt_[0] = mt_[N - 1]; i = 1; }
        }
        mt_[0] = 0x80000000u;
    }
    uint32_t next_u32()
    {
        static const uint32_t mag01[2] = {0u, 0x9908b0dfu};
        if (idx_ >= N) {
            int kk;
            for (kk = 0; kk < N - M; ++kk) {
                const uint32_t y = (mt_[kk] & 0x80000000u) | (mt_[kk + 1] & 0x7fffffffu);
                mt_[kk] = mt_[kk + M] ^ (y >> 1) ^ mag01[y & 1u];
            }
            for (; kk < N - 1; ++kk) {
                const uint32_t y = (mt_[kk] & 0x80000000u) | (mt_[kk + 1] & 0x7fffffffu);
                mt_[kk] = mt_[kk + (M - N)] ^ (y >> 1) ^ mag01[y & 1u];
            }
            const uint32_t y = (mt_[N - 1] & 0x80000000u) | (mt_[0] & 0x7fffffffu);
            mt_[N - 1] = mt_[M - 1] ^ (y >> 1) ^ mag01[y & 1u];
            idx_ = 0;
        }
        uint32_t y = mt_[idx_++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    uint64_t getrandbits(int k)                       // k <= 64 (the script never needs more than 32)
    {
        if (k <= 32) return next_u32() >> (32 - k);
        const uint64_t lo = next_u32();
        const uint64_t hi = next_u32() >> (64 - k);
        return (hi << 32) | lo;
    }
    uint64_t randbelow(uint64_t n)                    // Random._randbelow_with_getrandbits
    {
        if (!n) return 0;
        int k = 0;
        for (uint64_t t = n; t; t >>= 1) ++k;
        uint64_t r = getrandbits(k);
        while (r >= n) r = getrandbits(k);
        return r;
    }
};

// ---- Python's repr(float) / str(float) ------------------------------------------------------
std::string py_repr(double x)
{
    if (std::isnan(x)) return "nan";
    if (std::isinf(x)) return x > 0 ? "inf" : "-inf";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::scientific);   // shortest round-trip digits
    std::string s(buf, r.ptr);
    const size_t epos = s.find('e');
    std::string mant = s.substr(0, epos);
    const int exp10 = std::atoi(s.c_str() + epos + 1);
    bool neg = false;
    if (!mant.empty() && mant[0] == '-') { neg = true; mant.erase(0, 1); }
    std::string digits;
    for (char c : mant) if (c != '.') digits.push_back(c);
    std::string out;
    if (exp10 >= -4 && exp10 < 16) {                  // float_repr_style 'r': fixed in this decade range
        if (exp10 >= 0) {
            if ((int)digits.size() <= exp10 + 1) out = digits + std::string(exp10 + 1 - digits.size(), '0') + ".0";
            else out = digits.substr(0, exp10 + 1) + "." + digits.substr(exp10 + 1);
        } else {
            out = "0." + std::string(-exp10 - 1, '0') + digits;
        }
    } else {
        out = digits.substr(0, 1);
        if (digits.size() > 1) out += "." + digits.substr(1);
        char eb[16];
        std::snprintf(eb, sizeof eb, "e%c%02d", exp10 < 0 ? '-' : '+', std::abs(exp10));
        out += eb;
    }
    return neg ? "-" + out : out;
}

std::vector<std::string> split_ws(const std::string& s)
{
    std::istringstream is(s);
    std::vector<std::string> out;
    for (std::string t; is >> t;) out.push_back(t);
    return out;
}

bool starts_with(const std::string& s, const char* p) { return s.compare(0, std::strlen(p), p) == 0; }

int count_char(const std::string& s, char c)
{
    int n = 0;
    for (char x : s) n += x == c;
    return n;
}

void usage(const char* argv0)
{
    std::cerr << "usage: " << argv0 << " -i in.g2o [-o new.g2o] [-n outliers] [-g groupsize] [--information=...]"
                 " [--seed N] [-l|--local] [-p|--perfectMatch]" << std::endl;
}

}  // namespace

int main(int argc, char** argv)
{
    std::string in, out = "new.g2o", information;
    bool have_information = false, have_seed = false, local = false, perfect = false;
    long outliers = 100, groupsize = 1;
    uint64_t seed = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i], val;
        auto take = [&](const char* shortf, const char* longf) -> bool {
            if (a == shortf || a == longf) { if (i + 1 >= argc) return false; val = argv[++i]; return true; }
            const std::string pre = std::string(longf) + "=";
            if (starts_with(a, pre.c_str())) { val = a.substr(pre.size()); return true; }
            return false;
        };
        if (take("-i", "--in")) in = val;
        else if (take("-o", "--out")) out = val;
        else if (take("-n", "--outliers")) outliers = std::atol(val.c_str());
        else if (take("-g", "--groupsize")) groupsize = std::atol(val.c_str());
        else if (take("--information", "--information")) { information = val; have_information = !val.empty(); }
        else if (take("--seed", "--seed")) { seed = (uint64_t)std::llabs(std::atoll(val.c_str())); have_seed = true; }
        else if (a == "-l" || a == "--local") local = true;
        else if (a == "-p" || a == "--perfectMatch") perfect = true;
        else if (a == "-h" || a == "--help") { usage(argv[0]); return 0; }
        else { std::cerr << "unknown option " << a << std::endl; usage(argv[0]); return 2; }
    }
    // checkOptions (generateDataset.py:16-41)
    if (outliers < 0) { std::cout << "Number of outliers (--outliers) must be >=0." << std::endl; return 1; }
    if (groupsize < 0) { std::cout << "Groupsize (--groupsize) must be >=0." << std::endl; return 1; }
    if (in.empty()) { std::cout << "Dataset to read (--in) must be given." << std::endl; return 1; }
    if (have_information) {
        const int c = count_char(information, ',');
        if (c != 0 && c != 5 && c != 20) {
            std::cout << "Information matrix must be given in full upper-triangular form." << std::endl;
            return 1;
        }
    }
    PyRandom rnd;
    if (have_seed) rnd.seed(seed); else rnd.seed_from_entropy();

    // readDataset (:44-85)
    std::ifstream f(in);
    if (!f) { std::cerr << "cannot open " << in << std::endl; return 1; }
    std::vector<std::string> lines;
    for (std::string l; std::getline(f, l);) lines.push_back(l + "\n");
    int mode = 0;
    for (const auto& l : lines) {
        if (starts_with(l, "VERTEX_SE2")) { mode = 2; break; }
        if (starts_with(l, "VERTEX_SE3")) { mode = 3; break; }
    }
    if (!mode) { std::cout << "! Invalid mode. It must be either 2 or 3 but was None" << std::endl; return 1; }
    const char* vertexStr = mode == 2 ? "VERTEX_SE2" : "VERTEX_SE3:QUAT";
    const char* edgeStr = mode == 2 ? "EDGE_SE2" : "EDGE_SE3:QUAT";
    std::vector<std::string> vertices, edges;
    for (const auto& l : lines) {
        if (starts_with(l, vertexStr)) vertices.push_back(l);
        else if (starts_with(l, edgeStr)) edges.push_back(l);
    }

    // writeDataset (:106-250)
    if (have_information && count_char(information, ',') == 0) {
        char* end = nullptr;
        const double d = std::strtod(information.c_str(), &end);
        if (end == information.c_str() || *end) {
            std::cout << "! Invalid value for information matrix." << std::endl;
            return 1;
        }
        char b[512];
        if (mode == 2) std::snprintf(b, sizeof b, "%f,0,0,%f,0,%f", d, d, d);
        else std::snprintf(b, sizeof b, "%f,0,0,0,0,0,%f,0,0,0,0,%f,0,0,0,%f,0,0,%f,0,%f", d, d, d, d, d, d);
        information = b;
    } else if (have_information && count_char(information, ',') != (mode == 2 ? 5 : 20)) {
        std::cout << "! Invalid number of entries in information matrix." << std::endl;
        return 1;
    }
    std::ofstream o(out);
    if (!o) { std::cerr << "cannot write " << out << std::endl; return 1; }
    for (const auto& v : vertices) o << v;
    const long poseCount = (long)vertices.size();
    for (const auto& e : edges) {
        const auto el = split_ws(e);
        const bool odom = el.size() > 2 && std::atol(el[1].c_str()) == std::atol(el[2].c_str()) - 1;
        if (!odom && !have_information) {              // information of the first loop closure in the file
            const size_t k = mode == 2 ? 6 : 21;
            information.clear();
            for (size_t q = el.size() - k; q < el.size(); ++q) information += (q > el.size() - k ? " " : "") + el[q];
            have_information = true;
        }
        o << e;
    }
    if (outliers > 0 && !have_information) {
        std::cerr << "no loop closure edge to copy the information matrix from; give --information" << std::endl;
        return 1;
    }
    std::string info_str = information;
    for (char& c : info_str) if (c == ',') c = ' ';
    for (long n = 0; n < outliers; ++n) {
        long v1 = 0, v2 = 0;
        while (v1 == v2) {
            v1 = rnd.randint(0, poseCount - 1 - groupsize);
            if (!local) v2 = rnd.randint(0, poseCount - 1 - groupsize);
            else v2 = rnd.randint(v1, std::min(poseCount - 1 - groupsize, v1 + 20));
            if (v1 > v2) std::swap(v1, v2);
            if (v2 == v1 + 1) v2 = v1 + 2;
        }
        double x1, x2, x3, q0 = 1, q1 = 0, q2 = 0, q3 = 0;
        if (mode == 2) {
            x1 = rnd.gauss(0, 0.3); x2 = rnd.gauss(0, 0.3); x3 = rnd.gauss(0, 10 * M_PI / 180.0);
        } else {
            x1 = rnd.gauss(0, 0.3); x2 = rnd.gauss(0, 0.3); x3 = rnd.gauss(0, 0.3);
            const double sigma = 10.0 * M_PI / 180.0;
            const double roll = rnd.gauss(0, sigma), pitch = rnd.gauss(0, sigma), yaw = rnd.gauss(0, sigma);
            const double sy = sin_sep(yaw * 0.5), cy = cos_sep(yaw * 0.5), sp = sin_sep(pitch * 0.5),
                         cp = cos_sep(pitch * 0.5), sr = sin_sep(roll * 0.5), cr = cos_sep(roll * 0.5);
            q0 = cr * cp * cy + sr * sp * sy;          // euler_to_quat returns (w, x, y, z) ...
            q1 = sr * cp * cy - cr * sp * sy;
            q2 = cr * sp * cy + sr * cp * sy;
            q3 = cr * cp * sy - sr * sp * cy;
        }
        bool ints = false;
        if (perfect) { x1 = x2 = x3 = 0; q0 = 1; q1 = q2 = q3 = 0; ints = true; }   // Python ints: printed "0", "1"
        for (long g = 0; g < groupsize; ++g) {
            o << (mode == 2 ? "EDGE_SE2" : "EDGE_SE3:QUAT") << " " << v1 << " " << v2;
            const double m2[3] = {x1, x2, x3};
            for (double x : m2) o << " " << (ints ? std::to_string((long)x) : py_repr(x));
            if (mode == 3) {                           // ... and the script writes them in that order
                const double q[4] = {q0, q1, q2, q3};
                for (double x : q) o << " " << (ints ? std::to_string((long)x) : py_repr(x));
            }
            o << " " << info_str << "\n";
            ++v1; ++v2;
        }
    }
    o.close();
    std::cout << "Done." << std::endl;
    return 0;
}
