// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY PINNED through oracle/_ref (see smallmat.hpp header).
// Restatement of legkilo/src/core/slam/voxel_map.cc:22-427.
#include "oracle_voxel_map.hpp"

#include <cmath>

namespace lko {

int voxel_plane_id = 0;

// pcl/pcl_macros.h (PCL 1.8, third-party, absent here): #define DEG2RAD(x) ((x)*0.017453293)
#define LKO_DEG2RAD(x) ((x)*0.017453293)

// voxel_map.cc:22-40
void calcBodyCov(Vec3& pb, const float range_inc, const float degree_inc, Mat3& cov) {
    if (pb[2] == 0) pb[2] = 0.0001;
    float range = std::sqrt(pb[0] * pb[0] + pb[1] * pb[1] + pb[2] * pb[2]);
    float range_var = range_inc * range_inc;
    double sd = std::sin(LKO_DEG2RAD(degree_inc));
    double direction_var = sd * sd;  // pow(sin(.),2); 2x2 diag(direction_var, direction_var)
    Vec3 direction = pb;
    normalize(direction);
    Mat3 direction_hat = skew(direction);
    Vec3 base_vector1 = vec3(1, 1, -(direction[0] + direction[1]) / direction[2]);
    normalize(base_vector1);
    Vec3 base_vector2 = cross(base_vector1, direction);
    normalize(base_vector2);
    Mat<3, 2> N;
    for (int i = 0; i < 3; ++i) N(i, 0) = base_vector1[i], N(i, 1) = base_vector2[i];
    Mat<3, 2> A = ((double)range * direction_hat) * N;
    Mat3 c1 = (direction * (double)range_var) * direction.T();
    Mat3 c2 = (A * direction_var) * A.T();
    cov = c1 + c2;
}

VoxelOctoTree::VoxelOctoTree(int max_layer, int layer, int points_size_threshold, int max_points_num,
                             float planer_threshold)
    : layer_(layer),
      planer_threshold_(planer_threshold),
      points_size_threshold_(points_size_threshold),
      max_points_num_(max_points_num),
      max_layer_(max_layer) {
    octo_state_ = 0;
    new_points_ = 0;
    update_size_threshold_ = 5;  // voxel_map.h:158
    init_octo_ = false;
    update_enable_ = true;
    for (int i = 0; i < 8; i++) leaves_[i] = nullptr;
    voxel_center_[0] = voxel_center_[1] = voxel_center_[2] = 0.0;
    quater_length_ = 0;
    plane_ptr_ = new VoxelPlane;
}

VoxelOctoTree::~VoxelOctoTree() {
    for (int i = 0; i < 8; i++) delete leaves_[i];
    delete plane_ptr_;
}

// voxel_map.cc:42-117
void VoxelOctoTree::init_plane(const std::vector<pointWithVar>& points, VoxelPlane* plane) {
    plane->plane_var_ = Mat6::Zero();
    plane->covariance_ = Mat3::Zero();
    plane->center_ = Vec3::Zero();
    plane->normal_ = Vec3::Zero();
    plane->points_size_ = (int)points.size();
    plane->radius_ = 0;
    for (const auto& pv : points) {
        plane->covariance_ += pv.point_w * pv.point_w.T();
        plane->center_ += pv.point_w;
    }
    plane->center_ = plane->center_ / (double)plane->points_size_;
    plane->covariance_ = plane->covariance_ / (double)plane->points_size_ - plane->center_ * plane->center_.T();
    double evalsReal[3];
    Mat3 evecs;
    eig_sym3(plane->covariance_, evalsReal, evecs);  // EigenSolver<Matrix3d>, real parts (:55-59)
    int evalsMin = 0, evalsMax = 0;                   // minCoeff/maxCoeff: first index on ties (:60-62)
    for (int k = 1; k < 3; ++k) {
        if (evalsReal[k] < evalsReal[evalsMin]) evalsMin = k;
        if (evalsReal[k] > evalsReal[evalsMax]) evalsMax = k;
    }
    int evalsMid = 3 - evalsMin - evalsMax;
    if (evalsMid > 2) evalsMid = evalsMin;  // all-equal eigenvalues: out-of-range read in the reference (UB)
    auto col = [&](int k) { return vec3(evecs(0, k), evecs(1, k), evecs(2, k)); };
    double invn = 1.0 / plane->points_size_;
    Mat3 J_Q = Mat3::Zero();
    J_Q(0, 0) = J_Q(1, 1) = J_Q(2, 2) = invn;
    if (evalsReal[evalsMin] < planer_threshold_) {
        for (size_t i = 0; i < points.size(); i++) {
            Mat<6, 3> J;
            Mat3 F;
            for (int m = 0; m < 3; m++) {
                if (m != evalsMin) {
                    Mat<1, 3> lhs = (points[i].point_w - plane->center_).T() /
                                    ((plane->points_size_) * (evalsReal[evalsMin] - evalsReal[m]));
                    Mat3 rhs = col(m) * col(evalsMin).T() + col(evalsMin) * col(m).T();
                    Mat<1, 3> F_m = lhs * rhs;
                    for (int c = 0; c < 3; ++c) F(m, c) = F_m(0, c);
                } else {
                    for (int c = 0; c < 3; ++c) F(m, c) = 0;
                }
            }
            Mat3 top = evecs * F;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) J(r, c) = top(r, c), J(3 + r, c) = J_Q(r, c);
            plane->plane_var_ += (J * points[i].var) * J.T();
        }
        plane->normal_ = col(evalsMin);
        plane->y_normal_ = col(evalsMid);
        plane->x_normal_ = col(evalsMax);
        plane->min_eigen_value_ = evalsReal[evalsMin];
        plane->mid_eigen_value_ = evalsReal[evalsMid];
        plane->max_eigen_value_ = evalsReal[evalsMax];
        plane->radius_ = std::sqrt(evalsReal[evalsMax]);
        plane->d_ = -(plane->normal_[0] * plane->center_[0] + plane->normal_[1] * plane->center_[1] +
                      plane->normal_[2] * plane->center_[2]);
        plane->is_plane_ = true;
        plane->is_update_ = true;
        if (!plane->is_init_) {
            plane->id_ = voxel_plane_id;
            voxel_plane_id++;
            plane->is_init_ = true;
        }
    } else {
        plane->is_update_ = true;
        plane->is_plane_ = false;
    }
}

// voxel_map.cc:119-137
void VoxelOctoTree::init_octo_tree() {
    if (temp_points_.size() > (size_t)points_size_threshold_) {
        init_plane(temp_points_, plane_ptr_);
        if (plane_ptr_->is_plane_ == true) {
            octo_state_ = 0;
            if (temp_points_.size() > (size_t)max_points_num_) {
                update_enable_ = false;
                std::vector<pointWithVar>().swap(temp_points_);
                new_points_ = 0;
            }
        } else {
            octo_state_ = 1;
            cut_octo_tree();
        }
        init_octo_ = true;
        new_points_ = 0;
    }
}

// voxel_map.cc:139-183
void VoxelOctoTree::cut_octo_tree() {
    if (layer_ >= max_layer_) {
        octo_state_ = 0;
        return;
    }
    for (size_t i = 0; i < temp_points_.size(); i++) {
        int xyz[3] = {0, 0, 0};
        if (temp_points_[i].point_w[0] > voxel_center_[0]) xyz[0] = 1;
        if (temp_points_[i].point_w[1] > voxel_center_[1]) xyz[1] = 1;
        if (temp_points_[i].point_w[2] > voxel_center_[2]) xyz[2] = 1;
        int leafnum = 4 * xyz[0] + 2 * xyz[1] + xyz[2];
        if (leaves_[leafnum] == nullptr) {
            leaves_[leafnum] = new VoxelOctoTree(max_layer_, layer_ + 1, layer_init_num_[layer_ + 1], max_points_num_,
                                                 planer_threshold_);
            leaves_[leafnum]->layer_init_num_ = layer_init_num_;
            leaves_[leafnum]->voxel_center_[0] = voxel_center_[0] + (2 * xyz[0] - 1) * quater_length_;
            leaves_[leafnum]->voxel_center_[1] = voxel_center_[1] + (2 * xyz[1] - 1) * quater_length_;
            leaves_[leafnum]->voxel_center_[2] = voxel_center_[2] + (2 * xyz[2] - 1) * quater_length_;
            leaves_[leafnum]->quater_length_ = quater_length_ / 2;
        }
        leaves_[leafnum]->temp_points_.push_back(temp_points_[i]);
        leaves_[leafnum]->new_points_++;
    }
    for (unsigned i = 0; i < 8; i++) {
        if (leaves_[i] != nullptr) {
            if (leaves_[i]->temp_points_.size() > (size_t)leaves_[i]->points_size_threshold_) {
                init_plane(leaves_[i]->temp_points_, leaves_[i]->plane_ptr_);
                if (leaves_[i]->plane_ptr_->is_plane_) {
                    leaves_[i]->octo_state_ = 0;
                    if (leaves_[i]->temp_points_.size() > (size_t)leaves_[i]->max_points_num_) {
                        leaves_[i]->update_enable_ = false;
                        std::vector<pointWithVar>().swap(leaves_[i]->temp_points_);
                        new_points_ = 0;  // (sic) the PARENT's counter, voxel_map.cc:172
                    }
                } else {
                    leaves_[i]->octo_state_ = 1;
                    leaves_[i]->cut_octo_tree();
                }
                leaves_[i]->init_octo_ = true;
                leaves_[i]->new_points_ = 0;
            }
        }
    }
}

// voxel_map.cc:185-241
void VoxelOctoTree::UpdateOctoTree(const pointWithVar& pv) {
    if (!init_octo_) {
        new_points_++;
        temp_points_.push_back(pv);
        if (temp_points_.size() > (size_t)points_size_threshold_) { init_octo_tree(); }
    } else {
        if (plane_ptr_->is_plane_) {
            if (update_enable_) {
                new_points_++;
                temp_points_.push_back(pv);
                if (new_points_ > update_size_threshold_) {
                    init_plane(temp_points_, plane_ptr_);
                    new_points_ = 0;
                }
                if (temp_points_.size() >= (size_t)max_points_num_) {
                    update_enable_ = false;
                    std::vector<pointWithVar>().swap(temp_points_);
                    new_points_ = 0;
                }
            }
        } else {
            if (layer_ < max_layer_) {
                int xyz[3] = {0, 0, 0};
                if (pv.point_w[0] > voxel_center_[0]) xyz[0] = 1;
                if (pv.point_w[1] > voxel_center_[1]) xyz[1] = 1;
                if (pv.point_w[2] > voxel_center_[2]) xyz[2] = 1;
                int leafnum = 4 * xyz[0] + 2 * xyz[1] + xyz[2];
                if (leaves_[leafnum] != nullptr) {
                    leaves_[leafnum]->UpdateOctoTree(pv);
                } else {
                    leaves_[leafnum] = new VoxelOctoTree(max_layer_, layer_ + 1, layer_init_num_[layer_ + 1],
                                                         max_points_num_, planer_threshold_);
                    leaves_[leafnum]->layer_init_num_ = layer_init_num_;
                    leaves_[leafnum]->voxel_center_[0] = voxel_center_[0] + (2 * xyz[0] - 1) * quater_length_;
                    leaves_[leafnum]->voxel_center_[1] = voxel_center_[1] + (2 * xyz[1] - 1) * quater_length_;
                    leaves_[leafnum]->voxel_center_[2] = voxel_center_[2] + (2 * xyz[2] - 1) * quater_length_;
                    leaves_[leafnum]->quater_length_ = quater_length_ / 2;
                    leaves_[leafnum]->UpdateOctoTree(pv);
                }
            } else {
                if (update_enable_) {
                    new_points_++;
                    temp_points_.push_back(pv);
                    if (new_points_ > update_size_threshold_) {
                        init_plane(temp_points_, plane_ptr_);
                        new_points_ = 0;
                    }
                    if (temp_points_.size() > (size_t)max_points_num_) {
                        update_enable_ = false;
                        std::vector<pointWithVar>().swap(temp_points_);
                        new_points_ = 0;
                    }
                }
            }
        }
    }
}

VoxelMapManager::~VoxelMapManager() {
    // the reference never frees the trees (no manager dtor); the oracle does
    for (auto& kv : voxel_map_) delete kv.second;
}

// voxel_map.cc:552-569
bool VoxelMapManager::mapSliding() {
    if (norm(position_last_ - last_slide_position) < config_setting_.sliding_thresh) return false;
    last_slide_position = position_last_;
    Vec3i k = voxelKeyFloor(position_last_, config_setting_.max_voxel_size_);  // the DOUBLE voxel size here
    const int hm = config_setting_.half_map_size;
    clearMemOutOfMap(k[0] + hm, k[0] - hm, k[1] + hm, k[1] - hm, k[2] + hm, k[2] - hm);
    return true;
}

// voxel_map.cc:571-594: erase (and delete) every root voxel whose key lies strictly outside the box.
// Returns the number of deleted roots (the reference only counts them for a commented-out debug print).
int VoxelMapManager::clearMemOutOfMap(int x_max, int x_min, int y_max, int y_min, int z_max, int z_min) {
    int delete_voxel_cout = 0;
    for (auto it = voxel_map_.begin(); it != voxel_map_.end();) {
        const Vec3i& loc = it->first;
        bool should_remove =
            loc[0] > x_max || loc[0] < x_min || loc[1] > y_max || loc[1] < y_min || loc[2] > z_max || loc[2] < z_min;
        if (should_remove) {
            delete it->second;
            it = voxel_map_.erase(it);
            delete_voxel_cout++;
        } else {
            ++it;
        }
    }
    return delete_voxel_cout;
}

// voxel_map.cc:287-334
void VoxelMapManager::BuildVoxelMap(const Mat3 rot, const Mat3 rot_cov, const Mat3 pos_cov) {
    float voxel_size = config_setting_.max_voxel_size_;
    float planer_threshold = config_setting_.planner_threshold_;
    int max_layer = config_setting_.max_layer_;
    int max_points_num = config_setting_.max_points_num_;
    std::vector<int> layer_init_num = config_setting_.layer_init_num_;

    std::vector<pointWithVar> input_points;
    size_t n = feats_down_world_.size() / 3;
    input_points.reserve(n);
    for (size_t i = 0; i < n; i++) {
        pointWithVar pv;
        pv.point_w = vec3(feats_down_world_[3 * i], feats_down_world_[3 * i + 1], feats_down_world_[3 * i + 2]);
        Vec3 point_this = vec3(feats_down_body_[3 * i], feats_down_body_[3 * i + 1], feats_down_body_[3 * i + 2]);
        Mat3 var;
        calcBodyCov(point_this, config_setting_.dept_err_, config_setting_.beam_err_, var);
        Mat3 point_crossmat = skew(point_this);
        Mat3 rE = rot * extR_;
        var = (rE * var) * rE.T() + ((-point_crossmat) * rot_cov) * (-point_crossmat).T() + pos_cov;
        pv.var = var;
        input_points.push_back(pv);
    }

    for (size_t i = 0; i < input_points.size(); i++) {
        const pointWithVar p_v = input_points[i];
        Vec3i position = voxelKeyFloor(p_v.point_w, voxel_size);
        auto iter = voxel_map_.find(position);
        if (iter != voxel_map_.end()) {
            iter->second->temp_points_.push_back(p_v);
            iter->second->new_points_++;
        } else {
            VoxelOctoTree* octo_tree = new VoxelOctoTree(max_layer, 0, layer_init_num[0], max_points_num, planer_threshold);
            voxel_map_[position] = octo_tree;
            octo_tree->quater_length_ = voxel_size / 4;
            octo_tree->voxel_center_[0] = (0.5 + position[0]) * voxel_size;
            octo_tree->voxel_center_[1] = (0.5 + position[1]) * voxel_size;
            octo_tree->voxel_center_[2] = (0.5 + position[2]) * voxel_size;
            octo_tree->temp_points_.push_back(p_v);
            octo_tree->new_points_++;
            octo_tree->layer_init_num_ = layer_init_num;
        }
    }
    for (auto iter = voxel_map_.begin(); iter != voxel_map_.end(); ++iter) { iter->second->init_octo_tree(); }
}

// voxel_map.cc:336-361
void VoxelMapManager::UpdateVoxelMap(const std::vector<pointWithVar>& input_points) {
    float voxel_size = config_setting_.max_voxel_size_;
    float planer_threshold = config_setting_.planner_threshold_;
    int max_layer = config_setting_.max_layer_;
    int max_points_num = config_setting_.max_points_num_;
    std::vector<int> layer_init_num = config_setting_.layer_init_num_;
    size_t plsize = input_points.size();
    for (size_t i = 0; i < plsize; i++) {
        const pointWithVar& p_v = input_points[i];
        Vec3i position = voxelKeyFloor(p_v.point_w, voxel_size);
        auto iter = voxel_map_.find(position);
        if (iter != voxel_map_.end()) {
            iter->second->UpdateOctoTree(p_v);
        } else {
            VoxelOctoTree* octo_tree = new VoxelOctoTree(max_layer, 0, layer_init_num[0], max_points_num, planer_threshold);
            voxel_map_[position] = octo_tree;
            octo_tree->layer_init_num_ = layer_init_num;
            octo_tree->quater_length_ = voxel_size / 4;
            octo_tree->voxel_center_[0] = (0.5 + position[0]) * voxel_size;
            octo_tree->voxel_center_[1] = (0.5 + position[1]) * voxel_size;
            octo_tree->voxel_center_[2] = (0.5 + position[2]) * voxel_size;
            octo_tree->UpdateOctoTree(p_v);
        }
    }
}

// voxel_map.cc:363-427
void VoxelMapManager::build_single_residual(pointWithVar& pv, const VoxelOctoTree* current_octo,
                                            const int current_layer, bool& is_success, double& prob,
                                            PointToPlane& single_ptpl) {
    int max_layer = config_setting_.max_layer_;
    double sigma_num = config_setting_.sigma_num_;

    double radius_k = 3;
    Vec3 p_w = pv.point_w;
    if (current_octo->plane_ptr_->is_plane_) {
        VoxelPlane& plane = *current_octo->plane_ptr_;
        float dis_to_plane =
            std::fabs(plane.normal_[0] * p_w[0] + plane.normal_[1] * p_w[1] + plane.normal_[2] * p_w[2] + plane.d_);
        float dis_to_center = (plane.center_[0] - p_w[0]) * (plane.center_[0] - p_w[0]) +
                              (plane.center_[1] - p_w[1]) * (plane.center_[1] - p_w[1]) +
                              (plane.center_[2] - p_w[2]) * (plane.center_[2] - p_w[2]);
        float range_dis = std::sqrt(dis_to_center - dis_to_plane * dis_to_plane);
        if (gate_margin_probe_) {   // test diagnostic only, see the header
            const double thr = radius_k * plane.radius_;
            const double m = std::fabs((double)range_dis - thr) / thr;
            if (m < gate_margin_probe_[0]) gate_margin_probe_[0] = m;
        }

        if (range_dis <= radius_k * plane.radius_) {
            Mat<1, 6> J_nq;
            for (int c = 0; c < 3; ++c) J_nq(0, c) = p_w[c] - plane.center_[c], J_nq(0, 3 + c) = -plane.normal_[c];
            double sigma_l = ((J_nq * plane.plane_var_) * J_nq.T())(0, 0);
            sigma_l += ((plane.normal_.T() * pv.var) * plane.normal_)(0, 0);
            if (gate_margin_probe_) {
                const double thr = sigma_num * std::sqrt(sigma_l);
                const double m = std::fabs((double)dis_to_plane - thr) / thr;
                if (m < gate_margin_probe_[1]) gate_margin_probe_[1] = m;
            }
            if (dis_to_plane < sigma_num * std::sqrt(sigma_l)) {
                is_success = true;
                double this_prob = 1.0 / (std::sqrt(sigma_l)) * std::exp(-0.5 * dis_to_plane * dis_to_plane / sigma_l);
                if (this_prob > prob) {
                    prob = this_prob;
                    pv.normal = plane.normal_;
                    single_ptpl.body_cov_ = pv.body_var;
                    single_ptpl.point_b_ = pv.point_b;
                    single_ptpl.point_w_ = pv.point_w;
                    single_ptpl.plane_var_ = plane.plane_var_;
                    single_ptpl.normal_ = plane.normal_;
                    single_ptpl.center_ = plane.center_;
                    single_ptpl.d_ = plane.d_;
                    single_ptpl.layer_ = current_layer;
                    single_ptpl.dis_to_plane_ =
                        plane.normal_[0] * p_w[0] + plane.normal_[1] * p_w[1] + plane.normal_[2] * p_w[2] + plane.d_;
                    single_ptpl.point_crossmat_ = pv.point_crossmat;
                }
                return;
            } else {
                return;
            }
        } else {
            return;
        }
    } else {
        if (current_layer < max_layer) {
            for (size_t leafnum = 0; leafnum < 8; leafnum++) {
                if (current_octo->leaves_[leafnum] != nullptr) {
                    VoxelOctoTree* leaf_octo = current_octo->leaves_[leafnum];
                    build_single_residual(pv, leaf_octo, current_layer + 1, is_success, prob, single_ptpl);
                }
            }
            return;
        } else {
            return;
        }
    }
}

}  // namespace lko
