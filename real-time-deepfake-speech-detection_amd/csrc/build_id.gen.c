const char afx_build_id_str[] = "402930274202";
