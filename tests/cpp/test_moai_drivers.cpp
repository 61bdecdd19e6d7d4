// test_moai_drivers.cpp -- the reference's own weight-free drivers, run UNCHANGED on the device and checked.
//
// The reference's test.cpp (:18, :24, :30) calls batch_input_test(), ct_pt_matrix_mul_test() and ct_ct_matrix_mul_test()
// (include/test/matrix_mul/test_batch_encode_encrypt.hpp:4-101, test_ct_pt_matrix_mul.hpp:4-147,
// test_ct_ct_matrix_mul.hpp:4-209) before the 12-layer run; none of the three reads a weight file, and each prints the first
// and last five decoded slots of ten result ciphertexts for a human to read.  This translation unit includes the reference's
// include/include.hpp exactly as its test.cpp does (every MOAI header, the bootstrapping included, against the seal:: shim),
// makes the same three calls with std::cout captured, and asserts every printed slot against its closed form:
//   batch_input_test           N = 2^16, X[i][j][k] = j + 1, slot 256 j + i of every ciphertext  ->  first five 1, last five 128
//   ct_pt_matrix_mul_test      N = 2^15, X as above (128 inputs), W = 1/128 (768 x 64)           ->  768 (j + 1) / 128: 6 ... 768
//   ct_ct_matrix_mul_test      N = 2^15, X = 1, W[j][k] = 0.01 (j + 1) (128 x 64 each)
//        column packing  X W^T: ciphertext i holds the i-th diagonal, 0.64 ((j + i) mod 128 + 1) at token slot j
//                               -> first five 0.64 (i + 1); last five 0.64 ((127 + i) mod 128 + 1)
//        diagonal packing (X W^T) W: sum_m 0.64 m * 0.01 m = 0.0064 * 128 * 129 * 257 / 6 = 4526.4896 in every slot
// plus the chain indices the drivers print.  Tolerance: the products run at scale 2^40 on 40-bit primes, where the
// reference's own evaluator tests accept 0.5 (tests/seal/evaluator.cpp:2971-4293); here 1e-3 relative + 1e-4 absolute.
#include "include.hpp"

#include <regex>

static int g_fail = 0;
#define CHECK(cond)                                                        \
    do                                                                     \
    {                                                                      \
        if (!(cond))                                                       \
        {                                                                  \
            g_fail++;                                                      \
            printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
        }                                                                  \
    } while (0)

struct Printed
{
    int index;                 // the "<index>-th ciphertext" of the line (1-based, as printed)
    std::vector<double> first; // five slots before "..."
    std::vector<double> last;  // five slots after
};

// the "<i>-th ciphertext: a b c d e ... v w x y z" lines of a driver's output, in order
static std::vector<Printed> printed_slots(const std::string &text)
{
    std::vector<Printed> out;
    std::stringstream in(text);
    std::string line;
    const std::regex head("^([0-9]+)-th ciphertext: (.*)$");
    std::smatch m;
    while (std::getline(in, line))
    {
        if (!std::regex_match(line, m, head))
        {
            continue;
        }
        Printed p;
        p.index = std::stoi(m[1]);
        std::stringstream rest(m[2]);
        std::string tok;
        bool after = false;
        while (rest >> tok)
        {
            if (tok == "...")
            {
                after = true;
                continue;
            }
            (after ? p.last : p.first).push_back(std::stod(tok));
        }
        out.push_back(p);
    }
    return out;
}

static std::vector<size_t> printed_chain_indices(const std::string &text)
{
    std::vector<size_t> out;
    const std::regex pat("Modulus chain index for [^:]*: ([0-9]+)");
    for (auto it = std::sregex_iterator(text.begin(), text.end(), pat); it != std::sregex_iterator(); ++it)
    {
        out.push_back(std::stoul((*it)[1]));
    }
    return out;
}

static bool close_to(double got, double want)
{
    return std::fabs(got - want) <= 1e-3 * std::fabs(want) + 1e-4;
}

template <class F>
static std::string captured(F &&f)
{
    std::ostringstream sink;
    std::streambuf *keep = std::cout.rdbuf(sink.rdbuf());
    f();
    std::cout.rdbuf(keep);
    return sink.str();
}

template <class Want>
static void check_lines(const char *name, const std::vector<Printed> &lines, size_t begin, size_t count, Want &&want)
{
    double worst = 0;
    CHECK(lines.size() >= begin + count);
    for (size_t l = begin; l < begin + count && l < lines.size(); l++)
    {
        const Printed &p = lines[l];
        CHECK(p.first.size() == 5 && p.last.size() == 5);
        for (size_t s = 0; s < p.first.size(); s++)
        {
            const double w = want(p.index, false);
            worst = std::max(worst, std::fabs(p.first[s] - w));
            CHECK(close_to(p.first[s], w));
        }
        for (size_t s = 0; s < p.last.size(); s++)
        {
            const double w = want(p.index, true);
            worst = std::max(worst, std::fabs(p.last[s] - w));
            CHECK(close_to(p.last[s], w));
        }
    }
    printf("%-42s %zu printed ciphertexts, largest |printed slot - closed form| = %.3e\n", name, count, worst);
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    omp_set_num_threads(std::min(omp_get_max_threads(), 16));
    struct timeval t0, t1;
    auto seconds = [&]() { return t1.tv_sec - t0.tv_sec + (t1.tv_usec - t0.tv_usec) / 1e6; };

    // ---- test.cpp:18 ------------------------------------------------------------------------------------------------
    gettimeofday(&t0, NULL);
    const std::string a = captured([] { batch_input_test(); });
    gettimeofday(&t1, NULL);
    {
        const auto lines = printed_slots(a);
        CHECK(lines.size() == 10);
        check_lines("batch_input_test", lines, 0, lines.size(), [](int, bool last) { return last ? 128.0 : 1.0; });
        const auto idx = printed_chain_indices(a);
        CHECK(idx.size() == 1 && idx[0] == 2); // {60, 40, 40, 60}: fresh ciphertexts carry three data primes
        printf("  (%.1f s)\n", seconds());
    }
    // ---- test.cpp:24 ------------------------------------------------------------------------------------------------
    gettimeofday(&t0, NULL);
    const std::string b = captured([] { ct_pt_matrix_mul_test(); });
    gettimeofday(&t1, NULL);
    {
        const auto lines = printed_slots(b);
        CHECK(lines.size() == 10);
        check_lines("ct_pt_matrix_mul_test", lines, 0, lines.size(), [](int, bool last) { return last ? 768.0 : 6.0; });
        const auto idx = printed_chain_indices(b);
        CHECK(idx.size() == 2 && idx[0] == 2 && idx[1] == 1);
        printf("  (%.1f s)\n", seconds());
    }
    // ---- test.cpp:30 ------------------------------------------------------------------------------------------------
    gettimeofday(&t0, NULL);
    const std::string c = captured([] { ct_ct_matrix_mul_test(); });
    gettimeofday(&t1, NULL);
    {
        const size_t split = c.find("Task: test diag-packing");
        CHECK(split != std::string::npos);
        const auto col = printed_slots(c.substr(0, split)), diag = printed_slots(c.substr(split == std::string::npos ? 0 : split));
        CHECK(col.size() == 10 && diag.size() == 10);
        check_lines("ct_ct_matrix_mul_test, column packing", col, 0, col.size(), [](int index, bool last) {
            const int i = index - 1; // diagonal
            return 0.64 * (((last ? 127 : 0) + i) % 128 + 1);
        });
        check_lines("ct_ct_matrix_mul_test, diagonal packing", diag, 0, diag.size(), [](int, bool) { return 0.0064 * 128.0 * 129.0 * 257.0 / 6.0; });
        const auto idx = printed_chain_indices(c);
        // enc x 3, enc w 3, X W^T 2, then w and X W^T at 2, (X W^T) W at 1
        CHECK(idx.size() == 6 && idx[0] == 3 && idx[1] == 3 && idx[2] == 2 && idx[3] == 2 && idx[4] == 2 && idx[5] == 1);
        printf("  (%.1f s)\n", seconds());
    }
    if (g_fail)
    {
        printf("---- captured output of the three drivers ----\n%s\n%s\n%s\n", a.c_str(), b.c_str(), c.c_str());
    }
    else
    {
        printf("ALL PASS\n");
    }
    return g_fail ? 1 : 0;
}
