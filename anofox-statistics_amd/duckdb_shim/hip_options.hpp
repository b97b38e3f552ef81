// hip_options.hpp — the options argument of the regression aggregates as the reference's parser reads it, shared by the
// DuckDB glue files of this directory (fit_agg_hip.cpp, family_agg_hip.cpp).
#pragma once
#include <math.h>
#include <string.h>

#include <algorithm>
#include <cctype>
#include <initializer_list>
#include <utility>

#include "duckdb.hpp"

#include "anofox_stats_hip.h"

namespace duckdb {
namespace hip_glue {

enum class HipModel : uint8_t { OLS, RIDGE, WLS };

// ---- options: the keys, aliases, defaults and error texts of the reference's parser ----
// (RegressionMapOptions::ParseFromValue src/include/map_options_parser.cpp:637-750, ExtractBool :21-45,
//  ExtractSolverType / ExtractHcType / ExtractLambdaScaling :222-266, VisitOptionEntries :343-373 — STRUCT and MAP
//  literals, lower-cased keys, unknown keys ignored :798 — GetRegularizationStrength map_options_parser.hpp:265-270;
//  defaults ols_aggregate.cpp:48-52, ridge_aggregate.cpp:49-54, wls_aggregate.cpp:49-54)
struct HipFitOptions {
	bool fit_intercept = true;
	bool compute_inference = false;
	double confidence_level = 0.95;
	AnofoxSolverType solver = ANOFOX_SOLVER_SVD; // accepted; the GPU path has one solver (results agree within 1e-10)
	AnofoxHcType hc_type = ANOFOX_HC_NONE;
	double alpha = 1.0;
	AnofoxLambdaScaling lambda_scaling = ANOFOX_LAMBDA_SCALING_RAW;
	bool drop_y_zero_x = false; // null_policy = 'drop_y_zero_x': read by the predict aggregates only (ols_predict_aggregate.cpp:241-249)
	bool operator==(const HipFitOptions &o) const {
		return fit_intercept == o.fit_intercept && compute_inference == o.compute_inference && confidence_level == o.confidence_level &&
		       solver == o.solver && hc_type == o.hc_type && alpha == o.alpha && lambda_scaling == o.lambda_scaling && drop_y_zero_x == o.drop_y_zero_x;
	}
};

inline string Lower(string s) {
	std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
	return s;
}

inline bool ExtractBool(const Value &v) {
	switch (v.type().id()) {
	case LogicalTypeId::BOOLEAN: return BooleanValue::Get(v);
	case LogicalTypeId::INTEGER:
	case LogicalTypeId::BIGINT: return v.GetValue<int64_t>() != 0;
	case LogicalTypeId::DOUBLE: return v.GetValue<double>() != 0.0;
	default: throw InvalidInputException("Cannot convert value of type %s to boolean", v.type().ToString().c_str());
	}
}

template <class ENUM>
ENUM ExtractEnum(const Value &v, const char *what, const char *valid, std::initializer_list<std::pair<const char *, ENUM>> table) {
	const string s = Lower(v.type().id() == LogicalTypeId::VARCHAR ? StringValue::Get(v) : v.ToString());
	for (auto &e : table)
		if (s == e.first) return e.second;
	throw InvalidInputException("Invalid %s: '%s'. Valid values are %s", what, s.c_str(), valid);
}

inline void ApplyOption(const string &raw_key, const Value &v, HipFitOptions &o, bool &has_alpha, double &alpha, bool &has_lambda, double &lambda) {
	if (v.IsNull()) return;
	const string key = Lower(raw_key);
	if (key == "fit_intercept" || key == "intercept") o.fit_intercept = ExtractBool(v);
	else if (key == "compute_inference" || key == "inference") o.compute_inference = ExtractBool(v);
	else if (key == "confidence_level" || key == "confidence") o.confidence_level = v.GetValue<double>(); // no range check upstream
	else if (key == "alpha") { has_alpha = true; alpha = v.GetValue<double>(); }
	else if (key == "lambda") { has_lambda = true; lambda = v.GetValue<double>(); }
	else if (key == "solver")
		o.solver = ExtractEnum<AnofoxSolverType>(v, "solver", "'qr', 'svd', 'cholesky'",
		                                         {{"qr", ANOFOX_SOLVER_QR}, {"svd", ANOFOX_SOLVER_SVD}, {"cholesky", ANOFOX_SOLVER_CHOLESKY}});
	else if (key == "hc_type")
		o.hc_type = ExtractEnum<AnofoxHcType>(v, "hc_type", "'none', 'hc0', 'hc1', 'hc2', 'hc3'",
		                                      {{"none", ANOFOX_HC_NONE}, {"hc0", ANOFOX_HC_HC0}, {"hc1", ANOFOX_HC_HC1}, {"hc2", ANOFOX_HC_HC2}, {"hc3", ANOFOX_HC_HC3}});
	else if (key == "lambda_scaling")
		o.lambda_scaling = ExtractEnum<AnofoxLambdaScaling>(v, "lambda_scaling", "'raw', 'glmnet'",
		                                                    {{"raw", ANOFOX_LAMBDA_SCALING_RAW}, {"glmnet", ANOFOX_LAMBDA_SCALING_GLMNET}});
	else if (key == "null_policy") { // ExtractNullPolicy map_options_parser.cpp:80-93 (every aggregate validates it, the predict ones use it)
		const string s = Lower(v.type().id() == LogicalTypeId::VARCHAR ? StringValue::Get(v) : v.ToString());
		if (s == "drop") o.drop_y_zero_x = false;
		else if (s == "drop_y_zero_x") o.drop_y_zero_x = true;
		else throw InvalidInputException("Invalid null_policy: '%s'. Valid values are 'drop', 'drop_y_zero_x'", s.c_str());
	}
	// every other key: ignored, as upstream (the legacy {'full_output': true} of the reference's examples must bind)
}

inline void ParseHipFitOptions(const Value &v, HipFitOptions &o) {
	if (v.IsNull()) return;
	bool has_alpha = false, has_lambda = false;
	double alpha = 0.0, lambda = 0.0;
	if (v.type().id() == LogicalTypeId::STRUCT) {
		auto &kids = StructValue::GetChildren(v);
		for (idx_t i = 0; i < kids.size(); i++) ApplyOption(StructType::GetChildName(v.type(), i), kids[i], o, has_alpha, alpha, has_lambda, lambda);
	} else if (v.type().id() == LogicalTypeId::MAP) {
		for (auto &entry : MapValue::GetChildren(v)) { // a list of {key, value} structs
			auto &kv = StructValue::GetChildren(entry);
			if (kv.size() != 2 || kv[0].IsNull()) continue;
			ApplyOption(kv[0].type().id() == LogicalTypeId::VARCHAR ? StringValue::Get(kv[0]) : kv[0].ToString(), kv[1], o, has_alpha, alpha,
			            has_lambda, lambda);
		}
	} else {
		throw InvalidInputException("Options must be a MAP or STRUCT, got %s", v.type().ToString().c_str());
	}
	if (has_alpha) o.alpha = alpha; // alpha wins over lambda
	else if (has_lambda) o.alpha = lambda;
}

inline AnofoxHipBatchOptions MakeHipOptions(HipModel model, const HipFitOptions &o) {
	AnofoxHipBatchOptions b;
	memset(&b, 0, sizeof b);
	b.model = model == HipModel::OLS ? ANOFOX_HIP_MODEL_OLS : (model == HipModel::RIDGE ? ANOFOX_HIP_MODEL_RIDGE : ANOFOX_HIP_MODEL_WLS);
	b.fit_intercept = o.fit_intercept;
	b.compute_inference = o.compute_inference;
	b.confidence_level = o.confidence_level;
	b.solver = o.solver;
	// ridge: alpha and its scaling, no HC branch (ridge.rs:36-229); ols / wls: HC standard errors (ols.rs:209-245)
	b.alpha = model == HipModel::RIDGE ? o.alpha : 0.0;
	b.lambda_scaling = model == HipModel::RIDGE ? o.lambda_scaling : ANOFOX_LAMBDA_SCALING_RAW;
	b.hc_type = model == HipModel::RIDGE ? ANOFOX_HC_NONE : o.hc_type;
	return b;
}

} // namespace hip_glue
} // namespace duckdb
