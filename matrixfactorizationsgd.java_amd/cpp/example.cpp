// example.cpp -- the C++ host surface end to end: train on a small dense matrix, predict.
// Build (also done by __graft_entry__.build()):
//   g++ -std=c++17 -O2 example.cpp -L../lib -lmfsgd -Wl,-rpath,'$ORIGIN/../lib' -o ../lib/mfsgd_example
// Needs an MI355X to run (there is no CPU path); exits 2 with the library's message otherwise.
#include <cstdio>
#include <exception>

#include "MatrixFactorizationSGD.hpp"

int main() {
    const int U = 100, I = 80, k = 8;
    std::vector<int32_t> u, i;
    std::vector<float> r;
    for (int a = 0; a < U; ++a)
        for (int b = 0; b < I; ++b) {
            u.push_back(a);
            i.push_back(b);
            r.push_back(1.0f + (float)((a * 7 + b * 3) % 5));
        }
    try {
        MatrixFactorizationSGD mf(U, I, k, 0.01f, 0.05f, 42);
        const std::vector<double> rmse = mf.train(u, i, r, 5);
        for (size_t e = 0; e < rmse.size(); ++e) std::printf("epoch %zu rmse %.6f\n", e + 1, rmse[e]);
        std::printf("predict(3,4) = %.6f (rating %.1f)\n", mf.predict(3, 4), r[3 * I + 4]);
        // the distributed surface through the same host: ONE rank whose ring is an RCCL self-ring, two item partitions
        // (what rank g of an N-GPU job runs, with world = N and its own user range)
        MatrixFactorizationSGD dist(U, I, k, 0.01f, 0.05f, 42, /*device*/ 0, /*nParts*/ 2);
        const auto id = MatrixFactorizationSGD::distributedId();
        const std::vector<double> drmse = dist.trainDistributed(u, i, r, 3, /*rank*/ 0, /*world*/ 1, id, nullptr, /*userOffset*/ 0, /*usersTotal*/ U);
        for (size_t e = 0; e < drmse.size(); ++e) std::printf("dsgd epoch %zu rmse %.6f\n", e + 1, drmse[e]);
        const auto blocks = dist.itemBlocks();
        const auto st = dist.ringStats();
        std::printf("dsgd blocks %zu (partitions %d %d) trained %lld bytes_sent %lld\n", blocks.size(), blocks[0].first, blocks[1].first, st[0], st[3]);
        return rmse.back() < rmse.front() && drmse.back() < drmse.front() ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "mfsgd: %s\n", e.what());
        return 2;
    }
}
