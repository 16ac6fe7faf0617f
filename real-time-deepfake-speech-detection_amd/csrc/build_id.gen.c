const char afx_build_id_str[] = "d1d1e3fb9044";
