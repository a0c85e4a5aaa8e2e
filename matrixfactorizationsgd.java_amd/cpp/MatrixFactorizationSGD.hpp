// MatrixFactorizationSGD.hpp -- C++ host mirror of the Java surface over the C-ABI.
//
// The reference's toolchain (a JDK) is absent from this image, so the compiled
// host side above the C-ABI is C++ (the task's rule for compiled references);
// it mirrors java/MatrixFactorizationSGD.java method for method: same names,
// same argument meaning, a non-zero status becomes std::runtime_error where
// Java throws RuntimeException.  Header only; link against libmfsgd.so.
#pragma once

#include <array>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mfsgd.h"

class MatrixFactorizationSGD {
   public:
    // n_parts >= 2: one rank of a DSGD job (this rank's users, the GLOBAL item count, world * parts-per-rank item
    // partitions) -- Java: MatrixFactorizationSGD(users, items, k, lr, lambda, seed, device, nParts)
    MatrixFactorizationSGD(int users, int items, int k, float lr, float lambda, long long seed, int device = 0, int n_parts = 0)
        : users_(users), items_(items), k_(k), n_parts_(n_parts > 1 ? n_parts : 1), seed_(seed) {
        mfsgd_config cfg = {};
        cfg.n_users = users;
        cfg.n_items = items;
        cfg.k = k;
        cfg.lr = lr;
        cfg.lambda = lambda;
        cfg.device = device;
        cfg.n_parts = n_parts;
        const int rc = mfsgd_create(&cfg, &h_);
        if (rc != MFSGD_OK) throw std::runtime_error(std::string("mfsgd_create: ") + mfsgd_last_error(nullptr));
    }
    ~MatrixFactorizationSGD() { close(); }
    MatrixFactorizationSGD(const MatrixFactorizationSGD&) = delete;
    MatrixFactorizationSGD& operator=(const MatrixFactorizationSGD&) = delete;

    // double[] train(int[] u, int[] i, float[] r, int epochs): RMSE after each epoch
    std::vector<double> train(const std::vector<int32_t>& u, const std::vector<int32_t>& i, const std::vector<float>& r,
                              int epochs) {
        if (u.size() != i.size() || u.size() != r.size()) throw std::invalid_argument("length mismatch");
        check(mfsgd_set_ratings(h_, u.data(), i.data(), r.data(), (int64_t)u.size()));
        if (!initialised_) {
            check(mfsgd_init_factors(h_, seed_));
            initialised_ = true;
        }
        std::vector<double> rmse((size_t)epochs, 0.0);
        check(mfsgd_train(h_, epochs, rmse.data()));
        return rmse;
    }

    float predict(int u, int i) {
        float out = 0.f;
        const int32_t uu = u, ii = i;
        check(mfsgd_predict(h_, &uu, &ii, &out, 1));
        return out;
    }
    std::vector<float> predict(const std::vector<int32_t>& u, const std::vector<int32_t>& i) {
        if (u.size() != i.size()) throw std::invalid_argument("length mismatch");
        std::vector<float> out(u.size());
        check(mfsgd_predict(h_, u.data(), i.data(), out.data(), (int64_t)u.size()));
        return out;
    }

    // int[][] recommend(int[] users, int topN): best items per user, best first (row-major users x topN)
    std::pair<std::vector<int32_t>, std::vector<float>> recommend(const std::vector<int32_t>& users, int topn) {
        std::vector<int32_t> items(users.size() * (size_t)topn);
        std::vector<float> scores(users.size() * (size_t)topn);
        check(mfsgd_recommend(h_, users.data(), (int32_t)users.size(), topn, items.data(), scores.data()));
        return {std::move(items), std::move(scores)};
    }

    double rmse() {
        double out = 0.0;
        check(mfsgd_rmse(h_, &out));
        return out;
    }

    std::pair<std::vector<float>, std::vector<float>> factors() {
        std::vector<float> p((size_t)users_ * k_), q((size_t)items_ * k_);
        check(mfsgd_get_factors(h_, p.data(), q.data()));
        return {std::move(p), std::move(q)};
    }

    // ---- DSGD over the GPUs of one node: Java distributedId() / plan() / trainDistributed() / itemBlocks() ---------
    using RingId = std::array<unsigned char, MFSGD_DSGD_ID_BYTES>;
    static RingId distributedId() {
        RingId id{};
        if (mfsgd_dsgd_unique_id(id.data()) != MFSGD_OK) throw std::runtime_error(mfsgd_dsgd_last_error(nullptr));
        return id;
    }
    // {userBegin[nParts + 1], itemPart[items]}
    static std::pair<std::vector<int32_t>, std::vector<int32_t>> plan(const std::vector<int64_t>& deg_user,
                                                                       const std::vector<int64_t>& deg_item, int n_parts) {
        std::vector<int32_t> ub((size_t)n_parts + 1), ip(deg_item.size());
        if (mfsgd_dsgd_plan(deg_user.data(), deg_item.data(), (int32_t)deg_user.size(), (int32_t)deg_item.size(), n_parts, ub.data(),
                            ip.data()) != MFSGD_OK)
            throw std::invalid_argument("plan: bad argument");
        return {std::move(ub), std::move(ip)};
    }
    // double[] trainDistributed(u, i, r, epochs, rank, world, id, itemPart /*nullable*/, userOffset, usersTotal):
    // u are LOCAL user indices, i global item indices; collective; returns the global RMSE after each epoch
    std::vector<double> trainDistributed(const std::vector<int32_t>& u, const std::vector<int32_t>& i, const std::vector<float>& r,
                                         int epochs, int rank, int world, const RingId& id, const std::vector<int32_t>* item_part,
                                         long long user_offset, long long users_total) {
        if (n_parts_ < 2) throw std::logic_error("created without item partitions");
        if (u.size() != i.size() || u.size() != r.size()) throw std::invalid_argument("length mismatch");
        if (!ring_) {
            if (item_part) {
                if ((int)item_part->size() != items_) throw std::invalid_argument("itemPart must have one entry per item");
                check(mfsgd_set_item_partition(h_, item_part->data()));
            }
            check(mfsgd_set_ratings(h_, u.data(), i.data(), r.data(), (int64_t)u.size()));
            check(mfsgd_init_p_offset(h_, seed_, user_offset));
            if (mfsgd_dsgd_create(h_, rank, world, id.data(), &ring_) != MFSGD_OK) throw std::runtime_error(mfsgd_dsgd_last_error(nullptr));
            slots_ = n_parts_ / world;
            dcheck(mfsgd_dsgd_init_q(ring_, seed_, users_total));
            initialised_ = true;
        }
        std::vector<double> rmse((size_t)epochs, 0.0);
        dcheck(mfsgd_dsgd_train(ring_, epochs, rmse.data()));
        return rmse;
    }
    double rmseDistributed() {
        double out = 0.0;
        dcheck(mfsgd_dsgd_rmse(ring_, &out));
        return out;
    }
    std::vector<float> userFactors() {
        std::vector<float> p((size_t)users_ * k_);
        check(mfsgd_get_factors(h_, p.data(), nullptr));
        return p;
    }
    // the Q blocks held between epochs: {partition id, rows x k block} per slot
    std::vector<std::pair<int, std::vector<float>>> itemBlocks() {
        std::vector<std::pair<int, std::vector<float>>> out;
        for (int j = 0; j < slots_; ++j) {
            int32_t part = 0, rows = 0;
            dcheck(mfsgd_dsgd_get_q(ring_, j, &part, &rows, nullptr));
            std::vector<float> blk((size_t)rows * k_);
            dcheck(mfsgd_dsgd_get_q(ring_, j, &part, &rows, blk.data()));
            out.emplace_back(part, std::move(blk));
        }
        return out;
    }
    // {sub-epoch trainings, of those with the recovery point, of those re-run as round launches, bytes sent}
    std::array<long long, 4> ringStats() {
        int64_t s[4] = {0, 0, 0, 0};
        if (ring_) mfsgd_dsgd_stats(ring_, s);
        return {s[0], s[1], s[2], s[3]};
    }

    void close() {
        if (ring_) mfsgd_dsgd_destroy(ring_);
        ring_ = nullptr;
        if (h_) mfsgd_destroy(h_);
        h_ = nullptr;
    }

   private:
    void check(int rc) {
        if (rc != MFSGD_OK) throw std::runtime_error(mfsgd_last_error(h_));
    }
    void dcheck(int rc) {
        if (rc != MFSGD_OK) throw std::runtime_error(mfsgd_dsgd_last_error(ring_));
    }
    mfsgd_handle* h_ = nullptr;
    mfsgd_dsgd* ring_ = nullptr;
    int users_, items_, k_, n_parts_, slots_ = 0;
    long long seed_;
    bool initialised_ = false;
};
