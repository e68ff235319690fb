// nmpc_params_file.h -- host-only reader of ROS-2 parameter files shaped like the reference's
// config/mpc_params.yaml.
//
// The reference consumes that file twice: at code generation (acados/mpcPathTracking.py:83-92,
// 105-108,126: mpc_N, mpc_dt, wheelbase, the four *_range lists, nlp_solver_type -- a missing key is
// a KeyError there) and at run time as ROS parameters (src/RosMpc.cpp:20,34: path_tracking_vel,
// steering_ratio with default 1.0; loop rate config/mpc_params.yaml:9).  Here the same keys fill an
// nmpc_config at run time, so changing them needs no rebuild.
//
// Only the YAML subset such files use is understood: nested maps by indentation, `key: scalar`,
// `key: [a, b]` flow lists, `- item` block lists, quotes and `#` comments.  Keys are looked up under
// <node>/ros__parameters of the first node that has the mpc keys (the reference's path is
// "/mpc_local_planner"/"mpc_local_planner"/"ros__parameters"), or at top level for a flat file.
#pragma once

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <utility>
#include <vector>

namespace nmpcy {

typedef std::map<std::string, std::vector<std::string>> Table;   // "a/b/key" -> values

inline std::string trim(const std::string &s)
{
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r' || s[a] == '\n')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r' || s[b - 1] == '\n')) --b;
    return s.substr(a, b - a);
}

inline std::string strip_comment(const std::string &s)
{
    char q = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        if (q) {
            if (c == q) q = 0;
        } else if (c == '"' || c == '\'') {
            q = c;
        } else if (c == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) {
            return s.substr(0, i);
        }
    }
    return s;
}

inline std::string unquote(const std::string &s)
{
    if (s.size() >= 2 && (s[0] == '"' || s[0] == '\'') && s[s.size() - 1] == s[0]) return s.substr(1, s.size() - 2);
    return s;
}

inline std::vector<std::string> split_values(const std::string &v)
{
    std::vector<std::string> out;
    if (v.size() >= 2 && v[0] == '[' && v[v.size() - 1] == ']') {
        std::string item;
        const std::string body = v.substr(1, v.size() - 2);
        for (size_t i = 0; i <= body.size(); ++i) {
            if (i == body.size() || body[i] == ',') {
                const std::string t = trim(item);
                if (!t.empty()) out.push_back(unquote(t));
                item.clear();
            } else {
                item += body[i];
            }
        }
    } else {
        out.push_back(unquote(v));
    }
    return out;
}

// position of the ':' that ends a map key (followed by space or end of line, outside quotes), or npos
inline size_t key_colon(const std::string &s)
{
    char q = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        if (q) {
            if (c == q) q = 0;
        } else if (c == '"' || c == '\'') {
            q = c;
        } else if (c == ':' && (i + 1 == s.size() || s[i + 1] == ' ' || s[i + 1] == '\t')) {
            return i;
        }
    }
    return std::string::npos;
}

inline bool parse(FILE *f, Table &t, std::string &err)
{
    std::vector<std::pair<int, std::string>> stack;   // (indent, key) of the open maps
    std::string last_path;                            // key that a following block list belongs to
    char buf[1024];
    int lineno = 0;
    while (fgets(buf, sizeof(buf), f)) {
        ++lineno;
        const std::string raw = strip_comment(buf);
        const std::string body = trim(raw);
        if (body.empty() || body == "---") continue;
        int indent = 0;
        while (raw[indent] == ' ') ++indent;
        if (raw[indent] == '\t') {
            err = "line " + std::to_string(lineno) + ": tab indentation";
            return false;
        }
        if (body[0] == '-' && (body.size() == 1 || body[1] == ' ')) {
            if (last_path.empty()) {
                err = "line " + std::to_string(lineno) + ": list item without a key";
                return false;
            }
            const std::string item = trim(body.substr(1));
            if (!item.empty()) t[last_path].push_back(unquote(item));
            continue;
        }
        const size_t c = key_colon(body);
        if (c == std::string::npos) {
            err = "line " + std::to_string(lineno) + ": expected 'key: value'";
            return false;
        }
        while (!stack.empty() && stack.back().first >= indent) stack.pop_back();
        const std::string key = unquote(trim(body.substr(0, c)));
        const std::string val = trim(body.substr(c + 1));
        std::string path;
        for (size_t i = 0; i < stack.size(); ++i) path += stack[i].second + "/";
        path += key;
        if (val.empty()) {
            stack.push_back(std::make_pair(indent, key));
            t[path];                      // a map, or the head of a block list
            last_path = path;
        } else {
            t[path] = split_values(val);
            last_path.clear();
        }
    }
    return true;
}

inline bool to_double(const std::string &s, double *out)
{
    if (s.empty()) return false;
    char *end = nullptr;
    const double v = strtod(s.c_str(), &end);
    if (end == s.c_str() || *end != '\0') return false;
    *out = v;
    return true;
}

// prefix ("" or "a/b/") under which `key` lives; prefers .../ros__parameters/
inline bool find_prefix(const Table &t, const std::string &key, std::string *prefix)
{
    bool found = false;
    for (Table::const_iterator it = t.begin(); it != t.end(); ++it) {
        const std::string &p = it->first;
        const bool whole = p == key;
        const bool tail = p.size() > key.size() && p.compare(p.size() - key.size(), key.size(), key) == 0 &&
                          p[p.size() - key.size() - 1] == '/';
        if (!whole && !tail) continue;
        const std::string pre = p.substr(0, p.size() - key.size());
        const bool ros = pre.size() >= 16 && pre.compare(pre.size() - 16, 16, "ros__parameters/") == 0;
        if (!found || ros) {
            *prefix = pre;
            found = true;
            if (ros) return true;
        }
    }
    return found;
}

}  // namespace nmpcy
