// Host-side helpers shared by the multi-view back-end files (mvinit.hip, mvba.hip).
#pragma once
#include <string>
#include <vector>

namespace e2emv {
namespace mv {

struct Pair {
    int i = 0, j = 0;   // view ids, R_j = R_ij R_i
    double rot[3] = {0, 0, 0};  // angle-axis of R_ij
    double pos[3] = {0, 0, 0};  // position of camera j in the frame of camera i
};

std::vector<std::string> split_by_char(const std::string& s, char c);
void aa_to_R(const double* aa, double* R_colmajor);
void R_to_aa(const double* R_colmajor, double* aa);
bool estimate_rotations(int n_views, const std::vector<Pair>& pairs, std::vector<double>& rot);
bool estimate_positions(int n_views, const std::vector<Pair>& pairs, const std::vector<double>& rot, std::vector<double>& pos);

}  // namespace mv
}  // namespace e2emv
