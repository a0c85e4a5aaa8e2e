// MatrixFactorizationSGD.hpp -- C++ host mirror of the Java surface over the C-ABI.
//
// The reference's toolchain (a JDK) is absent from this image, so the compiled
// host side above the C-ABI is C++ (the task's rule for compiled references);
// it mirrors java/MatrixFactorizationSGD.java method for method: same names,
// same argument meaning, a non-zero status becomes std::runtime_error where
// Java throws RuntimeException.  Header only; link against libmfsgd.so.
#pragma once

#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mfsgd.h"

class MatrixFactorizationSGD {
   public:
    MatrixFactorizationSGD(int users, int items, int k, float lr, float lambda, long long seed, int device = 0)
        : users_(users), items_(items), k_(k), seed_(seed) {
        mfsgd_config cfg = {};
        cfg.n_users = users;
        cfg.n_items = items;
        cfg.k = k;
        cfg.lr = lr;
        cfg.lambda = lambda;
        cfg.device = device;
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

    void close() {
        if (h_) mfsgd_destroy(h_);
        h_ = nullptr;
    }

   private:
    void check(int rc) {
        if (rc != MFSGD_OK) throw std::runtime_error(mfsgd_last_error(h_));
    }
    mfsgd_handle* h_ = nullptr;
    int users_, items_, k_;
    long long seed_;
    bool initialised_ = false;
};
